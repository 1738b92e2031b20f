"""Inference fuzz: ``encoding`` of uint8 frames (value / 255 fused into the first layer, whatever its geometry) and of
the same frames as float32 on drawn architectures, both against the float64 CPU oracle (2e-5 of the largest latent).
    python tools/fuzz_encode.py          (seeds 40-63 on 1x64x64, 2x48x80, 1x128x128: 72 architectures)"""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from behavenet_amd.models import AE
from behavenet_amd.models.ae_model_architecture_generator import get_possible_arch
from oracle import ref_cpu
from tests.golden_utils import base_hparams
bad=0
for dim in ([1,64,64],[2,48,80],[1,128,128]):
    for seed in range(40,64):
        arch=get_possible_arch(list(dim),12,arch_seed=seed); arch.update(n_input_channels=dim[0],y_pixels=dim[1],x_pixels=dim[2])
        torch.manual_seed(0); hip=AE(base_hparams(dict(arch),'ae')).to('cuda').eval()
        torch.manual_seed(0); ora=ref_cpu.AE(base_hparams(dict(arch),'ae')).double().eval()
        g=torch.Generator().manual_seed(seed)
        xu=torch.randint(0,256,(7,)+tuple(dim),dtype=torch.uint8,generator=g)
        xf=(xu.numpy().astype(np.float32)/255)
        with torch.no_grad():
            zu=hip.encoding(xu.to('cuda'),dataset=0)[0].cpu().double()
            zf=hip.encoding(torch.from_numpy(xf).to('cuda'),dataset=0)[0].cpu().double()
            zo=ora.encoding(torch.from_numpy(xf).double(),dataset=0)[0]
        s=zo.abs().max().item()
        e1=(zu-zo).abs().max().item()/s; e2=(zf-zo).abs().max().item()/s; e3=(zu-zf).abs().max().item()/s
        ok = e1<=2e-5 and e2<=2e-5
        if not ok: bad+=1
        print('%s seed %d %s: u8 vs f64 %.1e, float vs f64 %.1e, u8 vs float %.1e'%('ok  ' if ok else 'FAIL',seed,dim,e1,e2,e3))
print('failures',bad)
