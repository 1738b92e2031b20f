#!/bin/bash
# A/B of the store policy of ONE producer launch inside the training step: the data gradient of
# enc.conv1 (k_up2_mfma<5, 4>, 134 MB of output) is read back by the very next kernel, the weight
# gradient of enc.conv0 (k_wgrad_c1d<true>).  Per library variant (UP2_ST_AUX_D = nt / plain / sc1):
# rocprofv3 kernel trace of the bench, the durations of that producer launch, of its consumer and
# their SUM.     usage: tools/ab_store_policy.sh   (needs ../libbn_auxd0.so, ../libbn_auxd16.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in nt:$R/behavenet_amd/libbehavenet_hip.so plain:$R/behavenet_amd/libbn_auxd0.so sc1:$R/behavenet_amd/libbn_auxd16.so nt_again:$R/behavenet_amd/libbehavenet_hip.so; do
  tag=${v%%:*}; lib=${v#*:}
  out=/tmp/abst_$tag; rm -rf $out
  BN_HIP_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d $out -o run -- \
      python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $out.log 2>&1
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
prod, cons, fwd = [], [], []
for i, (s, e, n) in enumerate(rows):
    if 'k_wgrad_c1d<true>' in n and i > 0 and 'k_up2_mfma<5, 4>' in rows[i - 1][2]:
        prod.append(rows[i - 1][1] - rows[i - 1][0]); cons.append(e - s)
    elif 'k_up2_mfma<5, 4>' in n and not (i + 1 < len(rows) and 'k_wgrad_c1d<true>' in rows[i + 1][2]):
        fwd.append(e - s)
k = len(prod) // 3          # steady state: last two thirds
med = lambda v: sorted(v)[len(v) // 2] / 1e3
print('%-9s pairs %3d | producer (enc.conv1 bwd-data) %6.1f us | consumer (enc.conv0 bwd-weight) %5.1f us | SUM %6.1f us | the other k_up2<5,4> launch (dec.convT3 fwd) %6.1f us' % (
    sys.argv[2], len(prod), med(prod[k:]), med(cons[k:]), med(prod[k:]) + med(cons[k:]), med(fwd[len(fwd) // 3:])))
PY
done
