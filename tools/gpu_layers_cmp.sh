#!/bin/bash
# per-layer conv table of one inner step: direct kernel only vs Winograd variants
mkdir -p gpurun_out
P2L_WINO=0 python tools/prof_layers.py > gpurun_out/layers_direct.txt 2>/dev/null
P2L_WINO_VARIANT=0 python tools/prof_layers.py > gpurun_out/layers_wino0.txt 2>/dev/null
P2L_WINO_VARIANT=1 python tools/prof_layers.py > gpurun_out/layers_wino1.txt 2>/dev/null
python tools/prof_layers.py > gpurun_out/layers_auto.txt 2>/dev/null; head -1 gpurun_out/layers_auto.txt
python - <<'PY'
def load(p):
    d = {}
    head = None
    for l in open(p):
        if '|' not in l or l.startswith('taps '):
            if l.startswith('candidates'): head = l.strip()
            continue
        a, b = l.split('|')
        d[tuple(a.split())] = b.split()
    return head, d
h0, d0 = load('gpurun_out/layers_direct.txt'); h1, d1 = load('gpurun_out/layers_wino0.txt'); h2, d2 = load('gpurun_out/layers_wino1.txt')
print(h0); print(h1); print(h2)
print('shape (taps B H W Cin Cout ups pro arb sk)            n   direct  wino0  wino1  ms/step')
tot = [0, 0, 0]
for k, v in sorted(d0.items(), key=lambda kv: -float(kv[1][1])):
    if k[0] != '9' or k[6] != '0': continue
    a, b, c = float(v[1]), float(d1.get(k, v)[1]), float(d2.get(k, v)[1])
    tot[0] += a; tot[1] += b; tot[2] += c
    print(' '.join('%5s' % x for x in k), '%5s' % v[0], '%7.3f %7.3f %7.3f' % (a, b, c), '  x%.2f x%.2f' % (a / b, a / c))
print('sum of plain 3x3 launches: direct %.2f  wino0 %.2f  wino1 %.2f ms/step' % tuple(tot))
PY
