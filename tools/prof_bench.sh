#!/bin/bash
# rocprofv3 kernel stats of the headline bench (no CPU baseline / secondary): per-kernel in-situ durations
# usage: tools/prof_bench.sh <tag> [env assignments...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $out/bench.log 2>&1 || true
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print('%-70s calls %5s avg %9.1f us  %5s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
