import sys, os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
from scratch.bench_shapes import *
import itertools
cf=[(0,0,0)]+[(t,n,k) for t in (3,4,8,9,6,7) for n in (2,3) for k in (16,21,28,32,42,56)]
for (M,N,c) in ((160,3072,2),(168,1024,3),(400,3072,2),(80,3072,2)):
    As=None
    wbytes=N*100352*2
    A=[torch.randn((M,100352),device=dev).bfloat16() for _ in range(2)]
    Ws=[(torch.randn((N,100352),device=dev)*0.05).bfloat16() for _ in range(c)]
    res=[]
    for cfg in cf:
        lib.trace_igemm_tune(*cfg)
        try: t=timed([(lambda i=i: ops.gemm_nt(A[i%2],Ws[i])) for i in range(c)])
        except RuntimeError as ex: t=float('nan')
        res.append((t,cfg))
    lib.trace_igemm_tune(0,0,0)
    d=res[0][0]
    res=[r for r in res if r[0]==r[0]]
    res.sort()
    print('M=%d N=%d: default %.1f us; best: '%(M,N,d)+'  '.join('%s %.1f(%.2fTB/s)'%(c_,t,wbytes/t/1e6) for t,c_ in res[:8]),flush=True)
    del A,Ws
