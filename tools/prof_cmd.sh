#!/bin/bash
# rocprofv3 kernel stats of any python command: tools/prof_cmd.sh <tag> <steps> python tools/...
set -e
tag=$1; steps=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- "$@" > $out/cmd.log 2>&1) || true
grep -v "^[EWI]2026" $out/cmd.log | tail -1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_kernel_stats.csv
python3 - "$f" "$steps" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step (%d steps incl. warm-up): %.3f ms' % (steps, tot / 1e6 / steps))
for r in rows[:22]:
    c = int(r['Calls']); avg = float(r['AverageNs']) / 1e3
    print('  %-66s per-step %5.2f avg %8.1f us  step-us %8.1f' % (r['Name'][:66], c / steps, avg, c * avg / steps))
PY
