# A/B of whole libraries inside the training step, every timed layer: tools/ab_layers.sh <lib.so> [<lib.so> ...]
for lib in "$@"; do BN_HIP_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-secondary --full-line 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); o=d['roofline_other_kernels']
print('%-34s %8.1f f/s %.3f ms | ' % ('$lib'.split('/')[-1], d['value'], d['ms_per_step']) + ' '.join('%s %.0f' % (e['layer'].split(' (')[0].replace('enc.','e').replace('dec.','d').replace('conv','c').replace(' bwd-weight','W').replace(' bwd-data','D').replace(' fwd','F'), e.get('avg_launch_us') or 0) for e in o if 'mfma' in (e.get('kernel') or '') or 'qg' in (e.get('kernel') or '')))"; done
