"""Same-box A/B of a model flag: python scratch/ab_flag.py <attr> <v0> <v1> ...  (values via eval)"""
import sys, os, time; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
import torch
from trace_amd.model_builder_rel import Generalized_RCNN
from trace_amd.synthetic import make_inputs, randomize_model
dev=torch.device('cuda',0)
attr=sys.argv[1]; vals=[eval(v) for v in sys.argv[2:]]
for rep in range(2):
  for val in vals:
    torch.manual_seed(1)
    m=Generalized_RCNN(); randomize_model(m,1); m.eval(); m.to(dev)
    m.Prd_RCNN.calibrate_bn((torch.randn((1,3,16,112,112))*50).to(dev))
    m.side_stream=True; m.graph_step=True; setattr(m,attr,val)
    frames,blob,feat,rois=make_inputs(8,32,20,224,seed=1)
    frames=frames.to(dev); blob=blob.to(dev).permute(0,2,3,1).contiguous().bfloat16(); feat=feat.to(dev).permute(0,2,3,1).contiguous().bfloat16()
    for _ in range(3): m(frames,blob,feat,rois,topk=100,prd_k=1)
    m.sync_steps()
    g=m._last_step_entry['graph']
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); print(attr,val,'graph ms %.4f'%((time.perf_counter()-t0)/50*1e3),flush=True)
    del m; torch.cuda.empty_cache()
