import sys, os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
import torch, numpy as np
from trace_amd import ops
from trace_amd.synthetic import make_inputs
dev='cuda:0'
frames,blob,feat,rois=make_inputs(8,8,20,224,seed=1)
r=torch.from_numpy(np.concatenate([np.concatenate([np.full((len(x),1),i,np.float32), x[:,1:]],1) for i,x in enumerate(rois)],0)).to(dev)
x=torch.randn((8,14,14,2048),device=dev).bfloat16()
def bench(f,n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    s.record(); g.replay(); e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
K=r.shape[0]
for sr in (2,0):
    t=bench(lambda: ops.roi_align_nhwc(x,r,1/16.,7,sr))
    by=K*49*2048*2+x.numel()*2+K*20
    print('roi_align_nhwc K=%d sr=%d: %.1f us  %.2f TB/s algorithmic'%(K,sr,t,by/t/1e6))
    t=bench(lambda: ops.roi_align_avg_nhwc(x,r,1/16.,7,sr))
    by=K*2048*2+x.numel()*2+K*20
    print('roi_align_avg_nhwc K=%d sr=%d: %.1f us  %.2f TB/s algorithmic'%(K,sr,t,by/t/1e6))
