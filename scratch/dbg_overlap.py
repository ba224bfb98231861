import sys, os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
import torch
from trace_amd.model_builder_rel import Generalized_RCNN
from trace_amd.synthetic import make_inputs, randomize_model
DEV=torch.device('cuda',0)
torch.manual_seed(1)
m=Generalized_RCNN(); randomize_model(m,1); m.eval(); m.to(DEV)
m.Prd_RCNN.calibrate_bn((torch.randn((1,3,16,112,112))*50).to(DEV))
a = make_inputs(2, 8, 6, 224, seed=21); b = make_inputs(2, 8, 6, 224, seed=22)
seq=[a,b,a,b,b,a]
keys=("prd_scores","msp_det_scores","det_labels")
def run(case):
    fr,bl,fe,ro=case
    return m(fr.to(DEV),bl.to(DEV),fe.to(DEV),ro,topk=50,prd_k=2)
def snap(outs):
    torch.cuda.synchronize(); return [{k:o[k].clone() for k in keys} for o in outs]
for mode in sys.argv[1:]:
    m.side_stream=True; m.graph_step=True; m.overlap_steps=False; m._step_graphs.clear(); m._ov_graphs.clear()
    m.split_pair_pool = "tail" if 'tail' in mode else False
    want=[snap(run(c)) for c in seq]
    want2=[snap(run(c)) for c in seq]
    print(mode,'want repeat diff',[max(float((x[k].float()-y[k].float()).abs().max()) for x,y in zip(g,w) for k in keys) for g,w in zip(want2,want)])
    m.overlap_steps=True
    got=[]
    for c in seq:
        o=run(c)
        if o is not None: got.append(snap(o))
    got.append(snap(m.flush_steps()))
    print(mode,'overlap diff',[max(float((x[k].float()-y[k].float()).abs().max()) for x,y in zip(g,w) for k in keys) for g,w in zip(got,want)])
