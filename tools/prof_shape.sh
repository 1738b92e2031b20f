#!/bin/bash
# rocprofv3 kernel stats of tools/step_shape.py: tools/prof_shape.sh <tag> C H W batch [steps rank world]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- \
    python $GRAFT_REPO_ROOT/tools/step_shape.py "$@" > $out/bench.log 2>&1 || true
tail -1 $out/bench.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step (30 steps): %.3f ms' % (tot / 1e6 / 30))
for r in rows[:36]:
    print('%-74s calls %5s avg %8.1f us %6.2f%%' % (r['Name'][:74], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
