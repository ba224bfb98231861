import sys, os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
from scratch.bench_shapes import *
for (M,N,c) in ((160,3072,2),(168,1024,3),(80,3072,2)):
    A=[torch.randn((M,100352),device=dev).bfloat16() for _ in range(2)]
    Ws=[(torch.randn((N,100352),device=dev)*0.05).bfloat16() for _ in range(c)]
    wbytes=N*100352*2
    res=[]
    for cfg in [(0,0,0),(10,0,8),(10,0,16),(10,0,21),(10,0,32),(8,3,21)]:
        lib.trace_igemm_tune(*cfg)
        t=timed([(lambda i=i: ops.gemm_nt(A[i%2],Ws[i])) for i in range(c)])
        res.append('%s %.1fus(%.2fTB/s)'%(cfg,t,wbytes/t/1e6))
    lib.trace_igemm_tune(0,0,0)
    print('M=%d N=%d: '%(M,N)+'  '.join(res),flush=True)
    del A,Ws
