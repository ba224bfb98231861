import sys, os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
from scratch.bench_shapes import gemm_case
cf=[(0,0,0)]+[(t,n,1) for t in (4,3,5,6,8,9,7) for n in (2,3,4)]
for M,N,K in ((6272,1536,1024),(25088,768,512),(6272,2048,512),(6272,2048,1024),(25088,512,128),(6272,1024,256),(6272,512,2048),(6272,512,1024),(6272,256,1024),(25088,256,512),(25088,128,512)):
    gemm_case(M,N,K,cf,copies=8)
