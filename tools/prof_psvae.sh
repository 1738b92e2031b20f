#!/bin/bash
# rocprofv3 kernel stats of the PS-VAE training step (BASELINE configs[3], tools/bench_psvae.py)
# usage: tools/prof_psvae.sh <tag> [env assignments...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- \
    python $GRAFT_REPO_ROOT/tools/bench_psvae.py 20 > $out/bench.log 2>&1 || true
cat $out/bench.log | tail -2
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step: %.3f ms' % (tot / 1e6 / 50))
for r in rows[:60]:
    print('%-70s calls %5s avg %9.1f us  %5s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
