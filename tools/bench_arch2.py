import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'behavenet_amd/configs/ae_jsons/ae_arch_2.json')
r = bench.geometry_step(p, [1, 128, 128], 'ae_arch_2.json on 1x128x128', batch=int(sys.argv[1]) if len(sys.argv) > 1 else 64)
# NB `dispatched_kernels`: four of this architecture's layers share the channel pair (64, 64), and the
# profiling hook selects by (family, channels) -- an entry there mixes those layers (name = the last one
# launched, time = their mean).  Per-layer truth: tools/probe_stride1.py, or rocprofv3 over tools/step_arch.py.
print(json.dumps(r, indent=1))
