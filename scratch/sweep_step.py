"""Collect every distinct conv / GEMM shape of one bench step, then sweep igemm configurations per
shape with cold operands (rotating copies larger than the caches).  Prints default vs best."""
import sys, os, itertools; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
import torch, numpy as np
from trace_amd import ops, _lib as L
from trace_amd.model_builder_rel import Generalized_RCNN
from trace_amd.synthetic import make_inputs, randomize_model
from scratch.bench_shapes import timed
dev=torch.device('cuda',0); lib=L.lib()
torch.manual_seed(1)
m=Generalized_RCNN(); randomize_model(m,1); m.eval(); m.to(dev)
frames,blob,feat,rois=make_inputs(8,32,20,224,seed=1)
frames=frames.to(dev); blob=blob.to(dev).permute(0,2,3,1).contiguous().bfloat16(); feat=feat.to(dev).permute(0,2,3,1).contiguous().bfloat16()
shapes={}
_g,_c=ops.gemm_nt,ops.conv3d
def g2(A,B,*a,**k):
    key=('g',tuple(A.shape),tuple(A.stride()),tuple(B.shape),tuple(B.stride()),bool(k.get('out_f32')),k.get('residual') is not None, k.get('out2'))
    shapes[key]=shapes.get(key,0)+1
    return _g(A,B,*a,**k)
def c2(x,w,b,ks,stride=(1,1,1),padding=(0,0,0),dilation=(1,1,1),**k):
    key=('c',tuple(x.shape),tuple(w.shape),tuple(ks),tuple(stride),tuple(padding),tuple(dilation),k.get('residual') is not None)
    shapes[key]=shapes.get(key,0)+1
    return _c(x,w,b,ks,stride,padding,dilation,**k)
ops.gemm_nt=g2; ops.conv3d=c2
import trace_amd.resnet3d as r3, trace_amd.model_builder_rel as mb
m(frames,blob,feat,rois,topk=100,prd_k=1); torch.cuda.synchronize()
ops.gemm_nt=_g; ops.conv3d=_c
print(len(shapes),'distinct shapes',flush=True)
tot_d=tot_b=0
for key,cnt in shapes.items():
    if key[0]=='g':
        _,ash,ast,bsh,bst,of32,res,o2=key
        nbytes=(np.prod(ash)+np.prod(bsh))*2
        copies=int(max(2,min(12,400e6//max(nbytes,1)+1)))
        As=[torch.randn(ash,device=dev).bfloat16() for _ in range(copies)]
        Bs=[(torch.randn(bsh,device=dev)*0.05).bfloat16() for _ in range(copies)]
        M=ash[-2]; N=bsh[-2]; K=ash[-1]; batch=max(ash[0] if len(ash)==3 else 1, bsh[0] if len(bsh)==3 else 1)
        fns=lambda: [(lambda i=i: _g(As[i],Bs[i],out_f32=of32)) for i in range(copies)]
        desc='GEMM b=%d M=%d N=%d K=%d'%(batch,M,N,K)
        fl=2.0*batch*M*N*K
    else:
        _,xsh,wsh,ks,st,pd,dl,res=key
        nbytes=(np.prod(xsh)+np.prod(wsh))*2
        copies=int(max(2,min(12,400e6//nbytes+1)))
        xs=[torch.randn(xsh,device=dev).bfloat16() for _ in range(copies)]
        ws=[(torch.randn(wsh,device=dev)*0.05).bfloat16() for _ in range(copies)]
        bb=torch.randn((wsh[0],),device=dev)
        y=_c(xs[0],ws[0],bb,ks,st,pd,dl,relu=True)
        M=y.numel()//wsh[0]; N=wsh[0]; K=wsh[1]; batch=1
        fns=lambda: [(lambda i=i: _c(xs[i],ws[i],bb,ks,st,pd,dl,relu=True)) for i in range(copies)]
        desc='CONV %s k%s s%s M=%d N=%d K=%d'%(xsh,ks,st,M,N,K)
        fl=2.0*M*N*K
    ksteps=(K+63)//64
    tiles=[1,2,3,4]+([8] if M<=256 else [])+([6] if N>=512 and M>=2048 else [])
    if N<=64: tiles=[1,2]
    kss=[1]
    if batch==1:
        kss+= [k for k in (2,3,4,6,8,12,16,21,32,48) if ksteps//k>=8]
    res_=[]
    lib.trace_igemm_tune(0,0,0); d=timed(fns())
    for t,n,k in itertools.product(tiles,(2,3,4),kss):
        stage={1:128,2:192,3:192,4:256,5:384,6:384,7:512,8:320,9:448}[t]*128
        if n*stage>160*1024: continue
        tc={1:64,2:64,3:128,4:128,6:256,8:128}[t]; tp={1:64,2:128,3:64,4:128,6:128,8:192}[t]
        nblk=-(-M//tp)*-(-N//tc)*batch*k
        if nblk>4096 and k>1: continue
        lib.trace_igemm_tune(t,n,k)
        try: tt=timed(fns())
        except RuntimeError: continue
        res_.append((tt,(t,n,k)))
    lib.trace_igemm_tune(0,0,0)
    res_.sort()
    tot_d+=d*cnt; tot_b+=res_[0][0]*cnt
    print('%-70s x%d default %6.1f us %5.0fTF | '%(desc,cnt,d,fl/d/1e6)+'  '.join('%s %.1f'%(c,t) for t,c in res_[:5]),flush=True)
    del fns
    try: del As,Bs
    except NameError: pass
    try: del xs,ws
    except NameError: pass
print('total default %.1f us, best %.1f us'%(tot_d,tot_b))
