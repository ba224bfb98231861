import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import torch, numpy as np
from trace_amd.model_builder_rel import Generalized_RCNN
from trace_amd.synthetic import make_inputs, randomize_model
dev=torch.device('cuda',0)
torch.manual_seed(1)
m=Generalized_RCNN(); randomize_model(m,1); m.eval(); m.to(dev)
m.Prd_RCNN.calibrate_bn((torch.randn((1,3,16,112,112))*50).to(dev))
frames,blob,feat,rois=make_inputs(8,32,20,224,seed=1)
frames=frames.to(dev); blob=blob.to(dev).permute(0,2,3,1).contiguous().bfloat16(); feat=feat.to(dev).permute(0,2,3,1).contiguous().bfloat16()
m.graph_step=True
for _ in range(3): outs=m(frames,blob,feat,rois,topk=100,prd_k=1)
frames,blob,feat=m.step_buffers()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(20): prep=m._host_prep(rois, 4, 1.0, 1.0)
t1=time.perf_counter(); print('host_prep ms',(t1-t0)/20*1e3)
t0=time.perf_counter()
for _ in range(20): prep['upl'].commit(dev)
torch.cuda.synchronize(); t1=time.perf_counter(); print('commit ms',(t1-t0)/20*1e3)
res=m._last_step_entry['res']
t0=time.perf_counter()
for _ in range(20): m._make_outs(prep,res,100,1)
t1=time.perf_counter(); print('make_outs ms',(t1-t0)/20*1e3)
g=m._last_step_entry['graph']
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); t1=time.perf_counter(); print('graph replay (GPU) ms',(t1-t0)/20*1e3)
t0=time.perf_counter()
for _ in range(20): outs=m(frames,blob,feat,rois,topk=100,prd_k=1)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter(); print('full step host ms',(t1-t0)/20*1e3,'incl sync',(t2-t0)/20*1e3)
