#!/bin/bash
# The round's rocprofv3 evidence with ONE library: tools/prof_round.sh <round tag, e.g. r04>
# (outputs under gpurun_out/; copy what is wanted into profiles/)
T=$1
R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_bench.sh ${T}_bench | tail -3
tools/prof_cmd.sh ${T}_ae_batchnorm 30 python tools/step_class.py ae bn 20 | head -3
tools/prof_cmd.sh ${T}_ae_arch2 30 python tools/step_arch.py behavenet_amd/configs/ae_jsons/ae_arch_2.json 1 128 128 256 20 | head -3
tools/prof_cmd.sh ${T}_1x64x48_b256 30 python tools/step_arch.py none 1 64 48 256 20 | head -3
tools/prof_cmd.sh ${T}_2x192x160_b256 30 python tools/step_arch.py none 2 192 160 256 20 | head -3
tools/prof_cmd.sh ${T}_1x192x192_b256 30 python tools/step_arch.py none 1 192 192 256 20 | head -3
tools/prof_cmd.sh ${T}_maxpool_arch 30 python tools/step_arch.py tests/golden/arch_maxpool.json 1 128 128 256 20 | head -3
tools/prof_cmd.sh ${T}_drawn_k3 30 python tools/step_arch.py tools/arch_jsons/drawn_k3.json 1 128 128 256 20 | head -3
tools/prof_cmd.sh ${T}_drawn_k7_k5_k9_k3 30 python tools/step_arch.py tools/arch_jsons/drawn_k7_k5_k9_k3.json 1 128 128 256 20 | head -3
tools/prof_cmd.sh ${T}_drawn_maxpool_k9_k7 30 python tools/step_arch.py tools/arch_jsons/drawn_maxpool_k9_k7.json 1 128 128 256 20 | head -3
BN_R=8 tools/prof_cmd.sh ${T}_frames_rank0of8 110 python tools/bench_frames_shard.py 40 | head -4   # 2 x (15 + 40) steps: eager, then the HIP graph; Adam on 1/8 of the arena
bash tools/prof_psvae.sh ${T}_psvae | head -3
for spec in "e0 E0 fwd" "e0w E0 bwd_w" "d4w D4 bwd_w" "d4l D4 fwd_sqerr"; do set -- $spec; bash tools/pmc_hbm.sh ${T}_$1 $2 $3; done
