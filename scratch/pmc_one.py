"""Runs one conv / GEMM shape a few times eagerly (for rocprofv3 --pmc passes)."""
import sys, os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
import torch
from trace_amd import ops, _lib as L
dev='cuda:0'
which=sys.argv[1]
cfg=tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv)>=5 else (0,0,0)
L.lib().trace_igemm_tune(*cfg)
B=8
if which=='l4_3x3':
    x=torch.randn((B,4,14,14,512),device=dev).bfloat16(); w=(torch.randn((512,9*512),device=dev)*0.05).bfloat16(); b=torch.randn((512,),device=dev)
    f=lambda: ops.conv3d(x,w,b,(1,3,3),(1,1,1),(0,2,2),(1,2,2),relu=True)
elif which=='l3_nl':
    x=torch.randn((B,4,14,14,1024),device=dev).bfloat16(); w=(torch.randn((512,1024),device=dev)*0.05).bfloat16(); b=torch.randn((512,),device=dev)
    f=lambda: ops.conv3d(x,w,b,(1,1,1),relu=True)
elif which=='l1_3x3':
    x=torch.randn((B,8,56,56,64),device=dev).bfloat16(); w=(torch.randn((64,9*64),device=dev)*0.05).bfloat16(); b=torch.randn((64,),device=dev)
    f=lambda: ops.conv3d(x,w,b,(1,3,3),(1,1,1),(0,1,1),(1,1,1),relu=True)
elif which=='fc1':
    x=torch.randn((160,100352),device=dev).bfloat16(); w=(torch.randn((3072,100352),device=dev)*0.05).bfloat16()
    f=lambda: ops.gemm_nt(x,w)
for _ in range(5): f()
torch.cuda.synchronize()
