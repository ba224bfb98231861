import sys, os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
import torch
from trace_amd import ops
dev='cuda:0'
N,T,H,W=8,32,224,224
clip=(torch.randn((N,T,3,H,W),device=dev)*50)
wp=(torch.randn((64,5*7*8*4),device=dev)*0.02).bfloat16(); b=torch.randn((64,),device=dev)
def bench(f,n=10):
    for _ in range(2): f()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
def old():
    xp=ops.clip_to_ndhwc(clip,True,4,(2,2,3,3,3,5)); return ops.conv3d_stem(xp,wp,b,5,7,8,(4,2,2),(8,112,112))
def new(): return ops.conv3d_stem_f32(clip,True,wp,b,4,(8,112,112))
assert torch.equal(old(),new())
print('convert + stem %.1f us   stem from f32 %.1f us'%(bench(old),bench(new)))
