# A/B of whole libraries inside the training step: tools/ab_lib.sh <lib.so> [<lib.so> ...] -- prints frames/s and rocprof-free step time
for lib in "$@"; do for rep in 1 2; do BN_HIP_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-secondary --full-line 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; o={e['layer']:e.get('avg_launch_us') for e in d['roofline_other_kernels']}
print('$lib', d['value'], d['ms_per_step'], 'conv0', r['avg_launch_us'], 'E1 fwd', o.get('enc.conv1 fwd'), 'E3 fwd', o.get('enc.conv3 fwd'), 'E1 bwd-d', o.get('enc.conv1 bwd-data'), 'D3 fwd', o.get('dec.convT3 fwd'))"; done; done
