#!/bin/bash
# SQ counter passes over the three MFMA kernel families (E2 layer, N=200).  Usage: tools/pmc_sq.sh <outdir>
set -e
out=${1:-gpurun_out/pmc_sq}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python tools/kbench.py --layers E2 --iters 3 --no-check"
rocprofv3 --kernel-trace --output-format csv -d $out/p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -- $CMD > $out.p1.log 2>&1 || true
rocprofv3 --kernel-trace --output-format csv -d $out/p2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE -- $CMD > $out.p2.log 2>&1 || true
rocprofv3 --kernel-trace --output-format csv -d $out/p3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM -- $CMD > $out.p3.log 2>&1 || true
find $out -name "*.csv" | head -20
