"""Per-MFMA-gap instruction census of a kernel's main loop in hipcc -S output.
usage: isa_gaps.py file.s kernel-name-substring [first-loop-label]"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith('_Z') and name in l and ':' in l and not l.startswith('\t'))
end = next(i for i in range(start, len(src)) if 's_endpgm' in src[i])
body = src[start:end]
# the main loop = from the first 'Loop Header' label to the last backward branch to it
hdr = next(i for i, l in enumerate(body) if 'Loop Header' in l)
label = body[hdr].split(':')[0]
last = max(i for i, l in enumerate(body) if re.search(r's_c?branch\S*\s+' + re.escape(label) + r'\b', l))
gap, tot, n = [], {}, 0
def cls(op):
    if op.startswith('v_mfma'): return 'MFMA'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'ds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'BARRIER'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_'): return 'salu'
    return None
for l in body[hdr:last + 1]:
    t = l.strip()
    if not t or t.startswith((';', '.')) or t.endswith(':'): continue
    c = cls(t.split()[0])
    if c is None: continue
    if c == 'MFMA':
        n += 1
        cnt = {}
        for g in gap: cnt[g] = cnt.get(g, 0) + 1
        print('%3d MFMA <- %2d: %s' % (n, len(gap), ' '.join('%s=%d' % kv for kv in sorted(cnt.items()))))
        gap = []
    else:
        gap.append(c); tot[c] = tot.get(c, 0) + 1
print('tail', len(gap), gap)
print('loop totals (non-MFMA):', tot, 'sum', sum(tot.values()), 'MFMAs', n)
