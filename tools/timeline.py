#!/usr/bin/env python
"""Per-step timeline from a rocprofv3 kernel trace: tools/timeline.py <kernel_trace.csv> [--all]

Steps are delimited by the optimizer kernel.  Prints wall / union-busy / summed kernel time of the
second-to-last step, the idle gaps > 20 us, and (with --all) every kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'k_adam' in r['Kernel_Name']]
i0, i1 = adam[-3], adam[-2]
step = rows[i0:i1 + 1]
t0 = int(step[0]['Start_Timestamp'])
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
busy, (cs, ce), gaps = 0, iv[0], []
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        if s - ce > 20000:
            gaps.append(((ce - t0) / 1e3, (s - ce) / 1e3))
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print('step wall %.1f us, busy %.1f us, summed kernel time %.1f us, %d kernels' % (
    (int(step[-1]['End_Timestamp']) - t0) / 1e3, busy / 1e3, sum(e - s for s, e in iv) / 1e3, len(step)))
print('idle gaps > 20 us (at, length):', ['%.0f:%.0f' % g for g in gaps])
if '--all' in sys.argv:
    for r in step:
        s = (int(r['Start_Timestamp']) - t0) / 1e3
        e = (int(r['End_Timestamp']) - t0) / 1e3
        print('%8.1f %8.1f %6.1f q%s %s' % (s, e, e - s, r.get('Queue_Id', '?'), r['Kernel_Name'][:60]))
