import sys, time, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import torch
from trace_amd.resnet3d import ResNet3d, i3d_dc5_config
torch.manual_seed(0)
m = ResNet3d(**i3d_dc5_config()); m.eval(); m.to('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
clip = torch.randn((B,32,3,224,224), device='cuda:0')*50
for _ in range(2): y = m.forward_ndhwc(clip, in_ntchw=True)
torch.cuda.synchronize()
n = 5
t = time.time()
for _ in range(n): y = m.forward_ndhwc(clip, in_ntchw=True)
torch.cuda.synchronize()
dt = (time.time()-t)/n
print('B', B, 'ms', dt*1e3, 'clips/s', B/dt, 'TFLOP/s', B*108.1e9/dt/1e12, y.shape)
