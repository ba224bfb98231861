import sys, os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
from scratch.bench_shapes import *
cf=[(0,0,0)]
print('NO_TCONV' if os.environ.get('TRACE_NO_TCONV') else 'TCONV')
conv_case('l1 64->64 3x1x1',64,8,56,56,64,(3,1,1),(1,1,1),(1,0,0),(1,1,1),cf)
conv_case('l1 256->64 3x1x1',256,8,56,56,64,(3,1,1),(1,1,1),(1,0,0),(1,1,1),cf)
conv_case('l2 256->128 3x1x1',256,4,56,56,128,(3,1,1),(1,1,1),(1,0,0),(1,1,1),cf)
conv_case('l2 512->128 3x1x1',512,4,28,28,128,(3,1,1),(1,1,1),(1,0,0),(1,1,1),cf)
conv_case('l3 512->256 3x1x1',512,4,28,28,256,(3,1,1),(1,1,1),(1,0,0),(1,1,1),cf)
conv_case('l3 1024->256 3x1x1',1024,4,14,14,256,(3,1,1),(1,1,1),(1,0,0),(1,1,1),cf)
conv_case('l4 2048->512 3x1x1',2048,4,14,14,512,(3,1,1),(1,1,1),(1,0,0),(1,1,1),cf)
