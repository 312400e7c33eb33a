"""from a rocprofv3 --kernel-trace CSV: launches with fewer than 256 workgroups, by total time"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    wg = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
    grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
    blocks = grid // max(wg, 1)
    if blocks < 256:
        k = (r['Kernel_Name'][:70], blocks)
        agg[k][0] += 1
        agg[k][1] += d
print('total kernel us', round(tot), ' underfilled (<256 blocks) us', round(sum(v[1] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%8.0f us %5d calls %7.1f us avg  blocks=%4d  %s' % (v[1], v[0], v[1] / v[0], k[1], k[0]))
