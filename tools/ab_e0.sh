# A/B of the enc.conv0 kernel generations INSIDE the training step (tuning build): tools/ab_e0.sh [variants...]
for v in ${@:-0 5 0 5}; do BN_HIP_LIB=$PWD/behavenet_amd/libbehavenet_hip_tuning.so BN_E0_V=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --full-line 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('BN_E0_V=$v', d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'])"; done
