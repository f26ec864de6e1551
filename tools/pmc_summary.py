#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel.

  python tools/pmc_summary.py OUT.json KERNEL_SUBSTRING DIR [DIR ...]

Each DIR is the -d directory of one `rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py ...` pass (FETCH_SIZE and
WRITE_SIZE are collected in separate passes, MI355X_MICROARCH.md HBM section).  Writes {counter: {avg_per_launch_kb |
avg_per_launch, launches}} for dispatches whose kernel name contains KERNEL_SUBSTRING; FETCH_SIZE / WRITE_SIZE are in KB.
"""
import csv, glob, json, os, sys

out, kern, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if kern not in r['Kernel_Name']:
                continue
            key = (r['Counter_Name'], r['Dispatch_Id'])
            per[key] = per.get(key, 0.0) + float(r['Counter_Value'])
        names = sorted({k[0] for k in per})
        for n in names:
            vals = [v for (c, _), v in per.items() if c == n]
            field = 'avg_per_launch_kb' if n in ('FETCH_SIZE', 'WRITE_SIZE') else 'avg_per_launch'
            res[n] = {field: sum(vals) / len(vals), 'launches': len(vals)}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
