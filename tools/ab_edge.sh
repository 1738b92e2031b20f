export BN_HIP_LIB=$PWD/behavenet_amd/libbehavenet_hip_tuning.so
for v in 0 1 2; do echo "== BN_E0_V=$v"; BN_E0_V=$v python tools/kbench.py --n 256 --layers E0,D4 --ops fwd,bwd_d --iters 50 --ring 3 2>&1 | grep -v "amdgpu.ids\|D4.*fwd"; done
