import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import torch, numpy as np
from trace_amd import ops, cluster_tree as ct
from conftest import synth_boxes
DEV='cuda:0'
n,G,F_,Hd=7,2,64,64
rng=np.random.default_rng(n); g=torch.Generator().manual_seed(n)
names=["Wz","Wr","Wh","Uz","Ur","Uh"]
w=[]
for gi in range(G):
    d={}
    for sfx in ("","d"):
        for nm in names:
            k=F_ if nm[0]=="W" else Hd
            d[nm+sfx]=(torch.randn((Hd,k),generator=g)/k**0.5).t().contiguous().to(DEV)
        d["bz"+sfx]=(torch.randn((Hd,),generator=g)*0.1).to(DEV); d["br"+sfx]=(torch.randn((Hd,),generator=g)*0.1).to(DEV)
    w.append(d)
trees=[ct.build_tree(synth_boxes(rng,nb)) for nb in (n,3)]
feats=[torch.randn((t["parent"].shape[0],G*F_),generator=g) for t in trees]
forest={k:torch.from_numpy(v).to(DEV) for k,v in ct.forest_arrays(trees).items()}
x=torch.cat(feats,0).to(DEV)
h0,hd0=ops.tree_gru(x,F_,Hd,G,forest,w,0)
h1,hd1=ops.tree_gru(x,F_,Hd,G,forest,w,max(t["parent"].shape[0] for t in trees))
torch.cuda.synchronize()
print('level_off', [t['level_off'] for t in trees])
for gi in range(G):
    dh=(h0-h1)[:,gi*Hd:(gi+1)*Hd].abs().max(1).values.cpu().numpy()
    dd=(hd0-hd1)[:,gi*Hd:(gi+1)*Hd].abs().max(1).values.cpu().numpy()
    print('g',gi,'h err per node',np.round(dh,4)); print('g',gi,'hd err per node',np.round(dd,4))
