import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from test_i3d_gpu import make_model
from oracle.i3d_ref import i3d_forward
from trace_amd import ops
m = make_model()
g = torch.Generator().manual_seed(8)
clip = torch.randn((1,3,8,64,64), generator=g)*50
ref, st = i3d_forward(m.state_dict(), clip, True)
m.to('cuda:0')
x = m._run_stem(clip.cuda(), False)
def err(a, r):
    a = a.float().permute(0,4,1,2,3).cpu()
    return float((a-r).norm()/r.norm()), float((a-r).abs().max()/r.abs().max())
print('stem', err(x, st[0]))
for i,name in enumerate(m.res_layers):
    for bi, blk in enumerate(getattr(m,name)):
        x = blk.run(x)
    if i==0: x = ops.maxpool3d(x,(2,1,1),(2,1,1))
    print(name, err(x, st[i+1]))
