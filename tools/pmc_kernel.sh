#!/bin/bash
# SQ counter passes over one kbench invocation.  Usage: tools/pmc_kernel.sh <outdir> <kbench args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python tools/kbench.py --iters 3 --no-check $@"
rocprofv3 --kernel-trace --output-format csv -d $out/p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -- $CMD > $out.p1.log 2>&1 || true
rocprofv3 --kernel-trace --output-format csv -d $out/p2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE -- $CMD > $out.p2.log 2>&1 || true
rocprofv3 --kernel-trace --output-format csv -d $out/p3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT -- $CMD > $out.p3.log 2>&1 || true
