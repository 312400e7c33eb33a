"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name:
calls, mean counter value per dispatch.  Usage: pmc_summary.py <csv> <out.csv>"""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
with open(src) as f:
    r = csv.DictReader(f)
    cols = r.fieldnames
    kn = [c for c in cols if 'Kernel_Name' in c][0]
    cn = [c for c in cols if 'Counter_Name' in c][0]
    cv = [c for c in cols if 'Counter_Value' in c][0]
    for row in r:
        key = (row[kn], row[cn])
        acc[key][0] += 1
        acc[key][1] += float(row[cv])
rows = sorted(((k[0], k[1], v[0], v[1], v[1] / v[0]) for k, v in acc.items()), key=lambda t: -t[3])
with open(dst, 'w') as f:
    w = csv.writer(f)
    w.writerow(['Kernel_Name', 'Counter', 'Dispatches', 'Sum', 'MeanPerDispatch'])
    for t in rows[:120]:
        w.writerow(t)
print(open(dst).read()[:1500])
