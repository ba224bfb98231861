import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
import oracle.i3d_ref as R
from test_i3d_gpu import make_model
m = make_model()
sd = {k: v.float() for k,v in m.state_dict().items()}
g = torch.Generator().manual_seed(8)
clip = torch.randn((1,3,8,64,64), generator=g)*50
ref, st = R.i3d_forward(sd, clip, True)
orig = R._conv_module
def rb(x): return x.bfloat16().float()
def run(round_act, round_w, fold=True):
    def cm(sd_, prefix, x, stride=1, padding=0, dilation=1, relu=True):
        import torch.nn.functional as F
        w = sd_[prefix+'.conv.weight']; b = sd_.get(prefix+'.conv.bias')
        if prefix+'.bn.weight' in sd_:
            scale = sd_[prefix+'.bn.weight']/torch.sqrt(sd_[prefix+'.bn.running_var']+1e-5)
            shift = sd_[prefix+'.bn.bias']-sd_[prefix+'.bn.running_mean']*scale
            w = w*scale.view(-1,1,1,1,1)
            if b is not None: shift = shift + b*scale
        else:
            shift = b
        if round_w: w = rb(w)
        y = F.conv3d(x, w, shift, stride=stride, padding=padding, dilation=dilation)
        if relu: y = F.relu(y)
        return rb(y) if round_act else y
    R._conv_module = cm
    try:
        x = rb(clip) if round_act else clip
        out, s2 = R.i3d_forward(sd, x, True)
    finally:
        R._conv_module = orig
    return [float((a-r).norm()/r.norm()) for a,r in zip(s2, st)]
print('act only ', run(True, False))
print('w only   ', run(False, True))
print('both     ', run(True, True))

def run2(which):
    def cm(sd_, prefix, x, stride=1, padding=0, dilation=1, relu=True):
        import torch.nn.functional as F
        y = orig(sd_, prefix, x, stride, padding, dilation, relu)
        inner = prefix.endswith('conv1') or prefix.endswith('conv2') and 'layer' in prefix
        nl = 'non_local' in prefix
        if which == 'inner' and inner and not nl: return rb(y)
        if which == 'nl' and nl: return rb(y)
        if which == 'outer' and not inner and not nl: return rb(y)
        return y
    R._conv_module = cm
    try:
        out, s2 = R.i3d_forward(sd, clip, True)
    finally:
        R._conv_module = orig
    return [round(float((a-r).norm()/r.norm()),4) for a,r in zip(s2, st)]
for w in ('inner','outer','nl'):
    print(w, run2(w))
