"""Summarise rocprofv3 --pmc counter_collection CSVs for one kernel -> json.
usage: python tools/pmc_summary.py <out.json> <kernel-substring> <csv> [<csv> ...]"""
import csv
import json
import sys
from collections import defaultdict

out, needle, files = sys.argv[1], sys.argv[2], sys.argv[3:]
vals = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if needle in r.get('Kernel_Name', ''):
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
res = {'kernel': needle, 'counters': {k: {'launches': len(v), 'mean': sum(v) / len(v), 'min': min(v), 'max': max(v)}
                                      for k, v in vals.items()}}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
