#!/usr/bin/env python
"""Aggregate rocprofv3 counter_collection.csv files per kernel: tools/pmc_show.py <dir> [filter]"""
import collections, csv, glob, sys
root = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
for f in sorted(glob.glob(root + '/*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in agg.items():
        if flt in k and k.lstrip('void ').startswith('k_'):
            print(k, {n: '%.3g' % (sum(v) / len(v)) for n, v in sorted(c.items())})
