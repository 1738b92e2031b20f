#!/usr/bin/env python
"""All idle gaps of the device in a rocprofv3 kernel trace: tools/gaps.py <kernel_trace.csv> [min_us]
(total wall, busy time, the gaps above the threshold with the kernels either side)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 200e3
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
ce, last = int(rows[0]['End_Timestamp']), rows[0]
busy, gaps = 0, []
cs = t0
for r in rows[1:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > ce:
        busy += ce - cs
        if s - ce > thr:
            gaps.append(((ce - t0) / 1e6, (s - ce) / 1e3, last['Kernel_Name'][:40], r['Kernel_Name'][:40]))
        cs, ce = s, e
    elif e > ce:
        ce = e
    if e >= ce:
        last = r
busy += ce - cs
wall = (ce - t0) / 1e6
print('wall %.1f ms, busy %.1f ms, idle %.1f ms in %d kernels' % (wall, busy / 1e6, wall - busy / 1e6, len(rows)))
tot = 0
for at, g, a, b in gaps:
    tot += g
    print('  at %8.1f ms: %8.1f us idle   after %-40s before %s' % (at, g, a, b))
print('gaps above %.0f us: %d, %.1f ms in total' % (thr / 1e3, len(gaps), tot / 1e3))
