#!/bin/bash
# Matrix-pipe utilisation of every kernel of the headline step from the SQ counters (separate --pmc passes):
#   tools/pmc_step.sh <tag>   ->  gpurun_out/pmc_<tag>_step/{p1,p2} + gpurun_out/<tag>_sq_step.txt
# SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs against GRBM_GUI_ACTIVE / 8 XCDs = share of the kernel's time its SIMDs'
# matrix pipes were busy (MI355X_MICROARCH.md, rocprofv3 section: per-XCD / per-SIMD aggregation of the counters).
tag=$1
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_step
rm -rf $out; mkdir -p $out
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-pmc"
BN_BENCH_PRIME=8 BN_BENCH_NOHOOK=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $out/p1 -o run -- $CMD > $out/p1.log 2>&1 || true
BN_BENCH_PRIME=8 BN_BENCH_NOHOOK=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out/p2 -o run -- $CMD > $out/p2.log 2>&1 || true
cd $GRAFT_REPO_ROOT
python - $out > gpurun_out/${tag}_sq_step.txt <<'PY'
import collections, csv, glob, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:64]][r['Counter_Name']].append(float(r['Counter_Value']))
print('%-66s %6s %10s %10s %8s %12s %12s' % ('kernel', 'calls', 'GUI_ACTIVE', 'MFMA_BUSY', 'mfma %', 'INSTS_MFMA', 'INSTS_VALU'))
rows = []
for k, c in agg.items():
    if not k.lstrip('void ').startswith('k_'):
        continue
    m = lambda n: (sum(c[n]) / len(c[n])) if c.get(n) else 0.0
    gui, busy = m('GRBM_GUI_ACTIVE'), m('SQ_VALU_MFMA_BUSY_CYCLES')
    util = (busy / 1024.0) / (gui / 8.0) if gui else 0.0
    rows.append((gui * len(c.get('GRBM_GUI_ACTIVE', [])), k, len(c.get('GRBM_GUI_ACTIVE', [])), gui, busy, util, m('SQ_INSTS_MFMA'), m('SQ_INSTS_VALU')))
for _, k, n, gui, busy, util, im, iv in sorted(rows, reverse=True)[:40]:
    print('%-66s %6d %10.3g %10.3g %7.1f%% %12.4g %12.4g' % (k, n, gui, busy, 100 * util, im, iv))
PY
head -30 gpurun_out/${tag}_sq_step.txt
