"""GPU-timed (graph-captured, no host overhead) sweep of igemm configs on plain GEMM shapes."""
import sys, os, itertools
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import torch
from trace_amd import ops, _lib as L
dev='cuda:0'
lib=L.lib()
def gtime(fn, n=20):
    fn(); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
shapes=[tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(4096,4096,4096),(6272,512,1024),(6272,256,2304),(25088,256,512)]
for M,N,K in shapes:
    A=torch.randn((M,K),device=dev).bfloat16(); B=(torch.randn((N,K),device=dev)*0.05).bfloat16()
    out=torch.empty((M,N),device=dev,dtype=torch.bfloat16)
    fl=2.0*M*N*K
    for tile,nst,ks in itertools.product((1,2,3,4),(2,3,4),(1,2,4)):
        lds={1:128,2:192,3:192,4:256}[tile]*128*nst
        if lds>160*1024: continue
        lib.trace_igemm_tune(tile,nst,ks)
        t=gtime(lambda: ops.gemm_nt(A,B,out=out))
        print('M%d N%d K%d tile=%d nst=%d ks=%d  %8.1f us %7.1f TF/s'%(M,N,K,tile,nst,ks,t,fl/t/1e6),flush=True)
    lib.trace_igemm_tune(0,0,0)
