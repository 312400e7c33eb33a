"""kernel time of ONE inner step by kernel name, from a rocprofv3 --kernel-trace CSV.

usage: python tools/step_kernels.py <kernel_trace.csv> [marker-substring]
The steps are delimited by the launches of a kernel that runs once per step (default: the forward
attention core); the long segments between two markers are full steps (forward +
loss + backward + Adam; the forward-only re-score passes of bench.py are shorter).  The trace's
timestamps of back-to-back dependent launches abut (a launch's start is stamped while its
predecessor drains), so idle time between kernels cannot be read from it: span == sum of durations."""
import csv, sys, collections
path = sys.argv[1]; marker = sys.argv[2] if len(sys.argv) > 2 else 'attn_core_kernel<0>'
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path)))
marks = [i for i, r in enumerate(rows) if marker in r[2]]
segs = [(b - a, a, b) for a, b in zip(marks, marks[1:])]
if not segs: sys.exit('fewer than 2 marker launches')
# (the first step also runs the one-off target preparation: take the LAST segment of the most common
#  launch count among the long ones)
big = [x for x in segs if x[0] >= 0.6 * max(segs)[0]]
mode = collections.Counter(x[0] for x in big).most_common(1)[0][0]
_, a, b = [x for x in big if x[0] == mode][-1]
seg = rows[a:b]
span = rows[b][0] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
print(f'step of {len(seg)} launches: span {span/1e6:.3f} ms, sum of kernel durations {busy/1e6:.3f} ms')
def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '').replace('p2lconv::', '')
    return n.split('(')[0][:70]
by = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    by[short(n)][0] += e - s; by[short(n)][1] += 1
print('   us/step  launches   us/launch  share  kernel')
for k, (g, c) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f'  {g/1e3:8.1f}  {c:8d}  {g/c/1e3:9.2f}  {100*g/busy:5.1f}  {k}')
