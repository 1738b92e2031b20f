#!/bin/bash
# rocprofv3 kernel durations of the qg_lab runs: tools/lab/qg_prof.sh <role> [N]
cd /tmp && export TMPDIR=/tmp
out=/tmp/qgprof_$1_$2; rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- $GRAFT_REPO_ROOT/tools/lab/bin/${QG_BIN:-qg_lab} $1 ${2:-256} 30 > $out.log 2>&1
grep -v "^\[" $out.log | grep -E "median|output" 
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print('   %-60s calls %4s avg %8.1f us min %8.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
