#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes as MI355X_MICROARCH.md prescribes) of one
# layer/op at 256 frames per launch.  usage: tools/pmc_hbm.sh <tag> <layer> <op>
tag=$1; layer=$2; op=$3
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o run -- \
      python $GRAFT_REPO_ROOT/tools/run_layer.py --layer $layer --op $op --n 256 --iters 6 > $out/$c.log 2>&1 || true
  f=$(find $out/$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c $tag <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if r.get('Counter_Name') == sys.argv[2]:
        acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k, v in acc.items():
    if len(v) >= 3 and ('k_' in k):
        print('%s %s %-60s launches %d  avg %.1f KB' % (sys.argv[3], sys.argv[2], k, len(v), sum(v[1:]) / len(v[1:])))
PY
done
