import sys, torch
sys.path.insert(0, '/root/repo')
from behavenet_amd import _hip
DEV='cuda'
def name_of(prof, fn):
    _hip.prof_select(prof, 0, 0)
    try:
        fn(); torch.cuda.synchronize()
        _, n, name = _hip.prof_read()
    finally:
        _hip.prof_select(_hip.PROF_NONE)
    return n, name
N=256
# conv2 bwd-data: conv 16->32 7x7 s1 on 64x64
geom=(N,16,64,64,32,7,7,1,3,3,64,64)
dy=torch.randn(N,32,64,64,device=DEV); w=torch.randn(32,16,7,7,device=DEV)
x=torch.randn(N,16,64,64,device=DEV)
print('conv2 bwd_d', name_of(_hip.PROF_CONV_BWD_D, lambda: _hip.conv2d_bwd_data(dy,w,geom,None,_hip.ACT_NONE,0.05)))
print('conv2 bwd_d mask', name_of(_hip.PROF_CONV_BWD_D, lambda: _hip.conv2d_bwd_data(dy,w,geom,x,_hip.ACT_LRELU,0.05)))
# convT1 fwd: 32->16 7x7 s1 64x64
geomT=(N,32,64,64,16,7,7,1,3,3,64,64)
xt=torch.randn(N,32,64,64,device=DEV); wt=torch.randn(32,16,7,7,device=DEV); bt=torch.randn(16,device=DEV)
print('convT1 fwd', name_of(_hip.PROF_CONVT_FWD, lambda: _hip.convT2d_fwd(xt,wt,bt,geomT,_hip.ACT_LRELU,0.05)))
geomT2=(N,16,128,128,1,9,9,1,4,4,128,128)
xt=torch.randn(N,16,128,128,device=DEV); wt=torch.randn(16,1,9,9,device=DEV); bt=torch.randn(1,device=DEV)
print('convT2 fwd', name_of(_hip.PROF_CONVT_FWD, lambda: _hip.convT2d_fwd(xt,wt,bt,geomT2,_hip.ACT_SIGMOID,0.05)))
