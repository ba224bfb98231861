"""Time every distinct conv/GEMM shape of I3D-R50-DC5 at B clips x 32f x 224^2."""
import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import torch
from trace_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev='cuda:0'
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
# (name, Cin, T,H,W, Cout, k, stride, pad, dil, count)
L = [
 ('l1 64->64 3x1x1', 64, 8,56,56, 64,(3,1,1),(1,1,1),(1,0,0),(1,1,1),1),
 ('l1 256->64 3x1x1',256, 8,56,56, 64,(3,1,1),(1,1,1),(1,0,0),(1,1,1),2),
 ('l1 64->64 1x3x3', 64, 8,56,56, 64,(1,3,3),(1,1,1),(0,1,1),(1,1,1),3),
 ('l1 64->256 1x1', 64, 8,56,56,256,(1,1,1),(1,1,1),(0,0,0),(1,1,1),4),
 ('l2 256->128 3x1x1',256,4,56,56,128,(3,1,1),(1,1,1),(1,0,0),(1,1,1),1),
 ('l2 128->128 3x3 s2',128,4,56,56,128,(1,3,3),(1,2,2),(0,1,1),(1,1,1),1),
 ('l2 256->512 1x1 s2',256,4,56,56,512,(1,1,1),(1,2,2),(0,0,0),(1,1,1),1),
 ('l2 128->512 1x1',128,4,28,28,512,(1,1,1),(1,1,1),(0,0,0),(1,1,1),4),
 ('l2 512->128 1x1',512,4,28,28,128,(1,1,1),(1,1,1),(0,0,0),(1,1,1),2),
 ('l2 512->128 3x1x1',512,4,28,28,128,(3,1,1),(1,1,1),(1,0,0),(1,1,1),1),
 ('l2 128->128 3x3',128,4,28,28,128,(1,3,3),(1,1,1),(0,1,1),(1,1,1),3),
 ('l2 NL 512->256 1x1',512,4,28,28,256,(1,1,1),(1,1,1),(0,0,0),(1,1,1),6),
 ('l2 NL 256->512 1x1',256,4,28,28,512,(1,1,1),(1,1,1),(0,0,0),(1,1,1),2),
 ('l3 512->256 3x1x1',512,4,28,28,256,(3,1,1),(1,1,1),(1,0,0),(1,1,1),1),
 ('l3 256->256 3x3 s2',256,4,28,28,256,(1,3,3),(1,2,2),(0,1,1),(1,1,1),1),
 ('l3 512->1024 1x1 s2',512,4,28,28,1024,(1,1,1),(1,2,2),(0,0,0),(1,1,1),1),
 ('l3 256->1024 1x1',256,4,14,14,1024,(1,1,1),(1,1,1),(0,0,0),(1,1,1),6),
 ('l3 1024->256 1x1',1024,4,14,14,256,(1,1,1),(1,1,1),(0,0,0),(1,1,1),3),
 ('l3 1024->256 3x1x1',1024,4,14,14,256,(3,1,1),(1,1,1),(1,0,0),(1,1,1),2),
 ('l3 256->256 3x3',256,4,14,14,256,(1,3,3),(1,1,1),(0,1,1),(1,1,1),5),
 ('l3 NL 1024->512 1x1',1024,4,14,14,512,(1,1,1),(1,1,1),(0,0,0),(1,1,1),10),
 ('l3 NL 512->1024 1x1',512,4,14,14,1024,(1,1,1),(1,1,1),(0,0,0),(1,1,1),3),
 ('l4 512->512 3x3 d2',512,4,14,14,512,(1,3,3),(1,1,1),(0,2,2),(1,2,2),3),
 ('l4 512->2048 1x1',512,4,14,14,2048,(1,1,1),(1,1,1),(0,0,0),(1,1,1),3),
 ('l4 1024->2048 1x1',1024,4,14,14,2048,(1,1,1),(1,1,1),(0,0,0),(1,1,1),1),
 ('l4 2048->512 3x1x1',2048,4,14,14,512,(3,1,1),(1,1,1),(1,0,0),(1,1,1),1),
 ('l4 2048->512 1x1',2048,4,14,14,512,(1,1,1),(1,1,1),(0,0,0),(1,1,1),1),
]
tot=0; totf=0
for name,cin,T,H,W,cout,k,s,p,d,cnt in L:
    x = torch.randn((B,T,H,W,cin),device=dev).bfloat16()
    w = (torch.randn((cout,k[0]*k[1]*k[2]*cin),device=dev)*0.05).bfloat16()
    b = torch.randn((cout,),device=dev)
    y = ops.conv3d(x,w,b,k,s,p,d,relu=True)
    ms = bench(lambda: ops.conv3d(x,w,b,k,s,p,d,relu=True))
    fl = 2.0*y.numel()*k[0]*k[1]*k[2]*cin
    by = 2.0*(x.numel()+y.numel()+w.numel())
    tmin = max(fl/2.5e15, by/8e12)*1e3
    print('%-22s M=%7d N=%5d K=%5d  %7.3f ms x%2d  %7.1f TF/s  %6.1f MB  roof %6.3f ms (%s) frac %.2f' % (name, y.numel()//cout, cout, k[0]*k[1]*k[2]*cin, ms, cnt, fl/ms/1e9, by/1e6, tmin, 'mfma' if fl/2.5e15>by/8e12 else 'hbm', tmin/ms))
    totmin = globals().get('totmin',0)+tmin*cnt
    tot += ms*cnt; totf += fl*cnt
# stem
xp = torch.randn((B,36,230,230,4),device=dev).bfloat16(); w=(torch.randn((64,5*7*8*4),device=dev)*0.05).bfloat16(); b=torch.randn((64,),device=dev)
ms = bench(lambda: ops.conv3d_stem(xp,w,b,5,7,8,(4,2,2),(8,112,112)))
fl = 2.0*B*8*112*112*64*735
print('%-22s %7.3f ms  %7.1f TF/s(useful) ' % ('stem st4', ms, fl/ms/1e9)); tot+=ms; totf+=fl
# NL gemms
for (nq,nkv,c,cnt) in ((3136,784,256,2),(784,196,512,3)):
    th = torch.randn((B,nq,c),device=dev).bfloat16(); ph=torch.randn((B,nkv,c),device=dev).bfloat16()
    ms1 = bench(lambda: ops.gemm_nt(th,ph,alpha=1.0/nkv))
    pw = ops.gemm_nt(th,ph,alpha=1.0/nkv); gt = torch.randn((B,c,nkv),device=dev).bfloat16()
    ms2 = bench(lambda: ops.gemm_nt(pw,gt))
    fl = 2.0*B*nq*nkv*c
    print('NL qk %dx%dx%d %7.3f ms %7.1f TF/s | pv %7.3f ms %7.1f TF/s  x%d' % (nq,nkv,c,ms1,fl/ms1/1e9,ms2,fl/ms2/1e9,cnt)); tot+=(ms1+ms2)*cnt; totf+=2*fl*cnt
print('roofline-min ms (convs only)', totmin); print('sum conv+gemm ms', tot, 'GFLOP', totf/1e9, 'avg TF/s', totf/tot/1e9)
