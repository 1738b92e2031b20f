import os, sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from behavenet_amd.models import AE
from behavenet_amd import hip_functions as hf
from behavenet_amd.models.ae_model_architecture_generator import get_possible_arch
from behavenet_amd.hostinfo import limit_host_threads
from oracle import ref_cpu
from tests.branches import record_branches, BranchReplay
from tests.golden_utils import base_hparams, make_frames
limit_host_threads(cap=32)
seed=410; n_frames=210; dim=[1,64,48]
arch=get_possible_arch(list(dim),12,arch_seed=seed); arch.update(n_input_channels=1,y_pixels=64,x_pixels=48)
extra={'ae_batch_norm':True}
torch.manual_seed(0); hip=AE(base_hparams(dict(arch),'ae',extra)).to('cuda')
torch.manual_seed(0); ora=ref_cpu.AE(base_hparams(dict(arch),'ae',extra)).double(); ora.train()
x=torch.from_numpy(make_frames(n_frames,dim,seed=500+seed))
kept=[]
orig=hf.BatchNormActFn.apply
def wrapped(xx,g,b,mod,act):
    xx.retain_grad(); out=orig(xx,g,b,mod,act); out.retain_grad(); kept.append((xx,out)); return out
hf.BatchNormActFn.apply=staticmethod(wrapped)
ograds={}; oin={}
def mk(name):
    def hook(mod,gin,gout): ograds.setdefault(name,[]).append((gin[0].detach().clone(),gout[0].detach().clone()))
    return hook
def mkf(name):
    def hook(mod,inp,out): oin.setdefault(name,[]).append((inp[0].detach().clone(), out.detach().clone()))
    return hook
for name,mod in ora.named_modules():
    if isinstance(mod,torch.nn.BatchNorm2d): mod.register_full_backward_hook(mk(name)); mod.register_forward_hook(mkf(name))
hip.train(); hip.zero_grad(set_to_none=True)
with record_branches(hip) as rec:
    lh=hip.loss({'images':x.to('cuda')[None]},dataset=0,accumulate_grad=True)['loss']
with BranchReplay(rec) as br:
    l64=ora.loss({'images':x.double()[None]},dataset=0,accumulate_grad=True)['loss']
name='encoding.encoder.batchnorm1'
xx,out=kept[1]
calls=ograds[name]
gin,gout=[c for c in calls if c[0].shape[0]==10][0]
xin,zout=[c for c in oin[name] if c[0].shape[0]==10][0]
dx_dev=xx.grad[200:210].cpu().double()
err=(dx_dev-gin).abs()
per_c=err.amax(dim=(0,2,3))
worst=per_c.argsort(descending=True)[:5].tolist()
var=xin.var(dim=(0,2,3),unbiased=False)
print('worst channels',worst,'their dx err',[float(per_c[c]) for c in worst],'max|dx|',float(gin.abs().max()))
print('their variance (f64, chunk 2):',[float(var[c]) for c in worst],' median var',float(var.median()))
xdev=xx.detach()[200:210].cpu().double()
print('x device vs oracle (chunk 2) max abs diff', float((xdev-xin).abs().max()), ' in worst channel', float((xdev[:,worst[0]]-xin[:,worst[0]]).abs().max()))
c=worst[0]
print('channel',c,'n bad elements',int((err[:,c]>1e-3*gin.abs().max()).sum()),'of',err[:,c].numel())
zdev=out.detach()[200:210,c].cpu().double(); 
print('branch disagreements in that channel between device y>0 and oracle z>0:', int(((zdev>0)!=(zout[:,c]>0)).sum()), ' min |z| oracle', float(zout[:,c].abs().min()))
