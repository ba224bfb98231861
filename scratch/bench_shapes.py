"""Cold-cache timing of single GEMM / conv shapes under igemm tune configurations.
Each timed graph holds R launches that rotate over R distinct weight (and input) copies whose total size
exceeds the Infinity Cache, so weights come from HBM as they do inside the real step.
usage: python scratch/bench_shapes.py"""
import sys, os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
import torch
from trace_amd import ops, _lib as L
dev='cuda:0'
lib=L.lib()

def timed(fns):
    side=torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for f in fns: f()
    torch.cuda.current_stream().wait_stream(side)
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns: f()
    g.replay(); torch.cuda.synchronize()
    best=1e9
    for _ in range(3):
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best=min(best,s.elapsed_time(e)/len(fns)*1e3)
    return best

def gemm_case(M,N,K,cfgs,copies=None,batch=1):
    wbytes=N*K*2*batch
    if copies is None: copies=max(2,min(64,int(600e6/wbytes)+1))
    As=[torch.randn((batch,M,K) if batch>1 else (M,K),device=dev).bfloat16() for _ in range(min(copies,8))]
    Ws=[(torch.randn((batch,N,K) if batch>1 else (N,K),device=dev)*0.05).bfloat16() for _ in range(copies)]
    out=[]
    for cfg in cfgs:
        lib.trace_igemm_tune(*cfg)
        try:
            t=timed([ (lambda i=i: ops.gemm_nt(As[i%len(As)],Ws[i])) for i in range(copies)])
        except RuntimeError as ex:
            t=float('nan')
        out.append((cfg,t))
    lib.trace_igemm_tune(0,0,0)
    fl=2.0*batch*M*N*K
    print('GEMM M=%d N=%d K=%d b=%d copies=%d (weights %.1f MB): '%(M,N,K,batch,copies,wbytes/1e6)+'  '.join('%s %.1fus(%.0fTF %.2fTB/s)'%(c,t,fl/t/1e6,wbytes/t/1e6) for c,t in out),flush=True)

def conv_case(name,cin,T,H,W,cout,k,s,p,d,cfgs,B=8,copies=None):
    K=k[0]*k[1]*k[2]*cin
    xbytes=B*T*H*W*cin*2
    if copies is None: copies=max(2,min(16,int(600e6/(xbytes+cout*K*2))+1))
    xs=[torch.randn((B,T,H,W,cin),device=dev).bfloat16() for _ in range(copies)]
    ws=[(torch.randn((cout,K),device=dev)*0.05).bfloat16() for _ in range(copies)]
    b=torch.randn((cout,),device=dev)
    y=ops.conv3d(xs[0],ws[0],b,k,s,p,d,relu=True)
    fl=2.0*y.numel()*K
    out=[]
    for cfg in cfgs:
        lib.trace_igemm_tune(*cfg)
        try:
            t=timed([(lambda i=i: ops.conv3d(xs[i],ws[i],b,k,s,p,d,relu=True)) for i in range(copies)])
        except RuntimeError:
            t=float('nan')
        out.append((cfg,t))
    lib.trace_igemm_tune(0,0,0)
    print('%s M=%d N=%d K=%d copies=%d: '%(name,y.numel()//cout,cout,K,copies)+'  '.join('%s %.1fus(%.0fTF)'%(c,t,fl/t/1e6) for c,t in out),flush=True)

if __name__=='__main__':
    which=sys.argv[1] if len(sys.argv)>1 else 'all'
    if which in ('small','all'):
        cf=[(0,0,0),(1,2,1),(1,3,1),(1,4,1),(1,4,2),(1,4,4),(3,4,1),(4,4,1)]
        gemm_case(160,1024,1024,cf)
        gemm_case(160,256,1024,cf)
        gemm_case(640,1024,2048,cf)
        gemm_case(3040,256,512,cf)
    if which in ('fc','all'):
        cf=[(0,0,0),(4,2,11),(4,2,21),(8,2,11),(8,2,10),(8,3,11),(9,2,21),(9,2,16),(5,2,11),(7,2,21)]
        gemm_case(160,3072,100352,cf,copies=2)
        gemm_case(168,1024,100352,cf,copies=3)
    if which in ('conv','all'):
        cf=[(0,0,0),(1,2,1),(1,3,1),(3,2,1),(3,3,1),(4,2,1),(4,3,1),(6,2,1),(2,2,1),(2,3,1)]
        conv_case('l4 3x3 d2',512,4,14,14,512,(1,3,3),(1,1,1),(0,2,2),(1,2,2),cf)
        conv_case('l3 3x3',256,4,14,14,256,(1,3,3),(1,1,1),(0,1,1),(1,1,1),cf)
        conv_case('l3 NL 1024->512',1024,4,14,14,512,(1,1,1),(1,1,1),(0,0,0),(1,1,1),cf)
        conv_case('l4 512->2048',512,4,14,14,2048,(1,1,1),(1,1,1),(0,0,0),(1,1,1),cf)
        conv_case('l2 3x3',128,4,28,28,128,(1,3,3),(1,1,1),(0,1,1),(1,1,1),cf)
        conv_case('l1 256->64 3x1x1',256,8,56,56,64,(3,1,1),(1,1,1),(1,0,0),(1,1,1),cf)
        conv_case('l1 64->64 3x3',64,8,56,56,64,(1,3,3),(1,1,1),(0,1,1),(1,1,1),cf)
