import sys, os, time, cProfile, pstats
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import torch, numpy as np
from trace_amd.model_builder_rel import Generalized_RCNN
from trace_amd.synthetic import make_inputs
m=Generalized_RCNN(); m.eval()
frames,blob,feat,rois=make_inputs(8,8,20,224,seed=1)
for _ in range(3): m._host_prep(rois,4,1.0,1.0)
t0=time.perf_counter()
for _ in range(20): m._host_prep(rois,4,1.0,1.0)
print('host_prep ms',(time.perf_counter()-t0)/20*1e3)
pr=cProfile.Profile(); pr.enable()
for _ in range(20): m._host_prep(rois,4,1.0,1.0)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
