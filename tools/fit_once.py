"""bench.py's fit() secondary on its own (a rocprofv3 / cProfile target): python tools/fit_once.py [epochs]
BN_FIT_ASYNC=0: checkpoints written synchronously (the reference's way); BN_FIT_NOSAVE=1: torch.save is a no-op
(what the file write costs the training loop)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from behavenet_amd import hip_functions as hf
hf.set_lazy_losses(True)
if os.environ.get('BN_FIT_NOSAVE') == '1':
    torch.save = lambda *a, **k: open(a[1], 'wb').close() if isinstance(a[1], str) else None
hp = bench.build_hparams()
if os.environ.get('BN_FIT_ASYNC') == '0':
    hp['async_checkpoint'] = False
out = bench.fit_throughput(hp, n_epochs=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
print(json.dumps({k: out[k] for k in ('value', 'seconds', 'ms_per_trial', 'trials_through_the_model')}))
