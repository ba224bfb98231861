import sys, os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
import numpy as np, torch
from trace_amd.model_builder_rel import Generalized_RCNN
from trace_amd.synthetic import make_inputs, randomize_model
from oracle.i3d_ref import calibrate_bn
from oracle.head_ref import head_forward
from oracle.np_oracle import topk_triplets
DEV='cuda:0'
torch.manual_seed(1)
m=Generalized_RCNN(); randomize_model(m,1)
calib=torch.randn((1,3,16,112,112),generator=torch.Generator().manual_seed(1))*50
m.Prd_RCNN.load_state_dict(calibrate_bn(m.Prd_RCNN.state_dict(),calib)); m.eval()
frames,blob,feat,rois=make_inputs(1,64,40,224,seed=21)
sd={k:v.detach().cpu() for k,v in m.state_dict().items()}
m.to(DEV)
outs=m(frames.to(DEV),blob.to(DEV),feat.to(DEV),rois,topk=100,prd_k=2)
tm=m._debug['tmaps'].float().permute(0,1,4,2,3).cpu()
ref=head_forward(sd,blob[0:1],feat[0:1],tm[0],rois[0])
rs=ref['det_scores']
sc,pair,prd=topk_triplets(ref['prd_scores'].numpy(),rs[ref['sbj_inds']],rs[ref['obj_inds']],2,110)
print('ref scores 90..110',sc[90:110])
o=outs[0]
print('got scores 90..100',o['top_scores'].cpu().numpy()[90:100])
want=set(zip(pair[:100].tolist(),prd[:100].tolist())); got=set(zip(o['top_pair'].cpu().tolist(),o['top_prd'].cpu().tolist()))
print('missing',want-got,'extra',got-want)
g=o['prd_scores'].cpu().numpy(); r=ref['prd_scores'].numpy()
print('max abs prd err',np.abs(g-r).max())
for (p_,k_) in list(want-got)+list(got-want):
    print(p_,k_,'ref',r[p_,k_]*rs[ref['sbj_inds'][p_]]*rs[ref['obj_inds'][p_]],'got',g[p_,k_]*float(o['sbj_scores'][p_])*float(o['obj_scores'][p_]))
