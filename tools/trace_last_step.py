"""One step out of a rocprofv3 --kernel-trace of tools/step_sg2_one.py (or any program whose steps end in
adam_dev_kernel launches): kernels between the last two optimiser updates, by name, with the idle time between
dispatches.  usage: trace_last_step.py <kernel_trace.csv> [out.txt]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
name_k = 'Kernel_Name' if 'Kernel_Name' in rows[0] else 'Name'
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r[name_k]) for r in rows))
adam = [i for i, e in enumerate(ev) if 'adam_dev_kernel' in e[2]]
# groups of consecutive adam launches (one group per step)
groups = []
for i in adam:
    if groups and i - groups[-1][-1] <= 3:
        groups[-1].append(i)
    else:
        groups.append([i])
lo, hi = groups[-2][-1] + 1, groups[-1][-1] + 1
step = ev[lo:hi]
wall = (step[-1][1] - step[0][0]) / 1e3
busy = 0.0
cur_end = step[0][0]
for s, e, _ in step:
    if e > cur_end:
        busy += (e - max(s, cur_end)) / 1e3
        cur_end = e
agg = collections.OrderedDict()
for s, e, n in step:
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = n.split('(')[0] if not n.startswith('at::') else n[:70]
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
out = ['last step: %d dispatches, %.1f us wall, %.1f us GPU busy (%.1f %% idle between dispatches)' % (
    len(step), wall, busy, 100 * (1 - busy / wall))]
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.append('%-90s %5d %10.1f us %5.1f%%' % (n[:90], c, t, 100 * t / busy))
txt = '\n'.join(out) + '\n'
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(txt)
print(txt)
