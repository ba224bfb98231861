import sys, os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_)
from scratch.bench_shapes import conv_case
cf=[(0,0,0)]
for cp in (2,16):
    conv_case('l1 64->64 3x3',64,8,56,56,64,(1,3,3),(1,1,1),(0,1,1),(1,1,1),cf,copies=cp)
    conv_case('l2 128 3x3',128,4,28,28,128,(1,3,3),(1,1,1),(0,1,1),(1,1,1),cf,copies=cp)
    conv_case('l3 256 3x3',256,4,14,14,256,(1,3,3),(1,1,1),(0,1,1),(1,1,1),cf,copies=cp)
    conv_case('l4 512 3x3 d2',512,4,14,14,512,(1,3,3),(1,1,1),(0,2,2),(1,2,2),cf,copies=cp)
