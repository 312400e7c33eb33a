#!/bin/bash
# per-layer conv table of one inner step under two settings of an environment switch:
#   bash tools/gpu_layers_cmp.sh P2L_PW 0 1 [taps]
mkdir -p gpurun_out
V=$1; A=$2; B=$3; T=${4:-1}
env $V=$A python tools/prof_layers.py > gpurun_out/layers_a.txt 2>/dev/null
env $V=$B python tools/prof_layers.py > gpurun_out/layers_b.txt 2>/dev/null
python - $V $A $B $T <<'PY'
import sys
V, A, B, T = sys.argv[1:5]
def load(p):
    d, head = {}, None
    for l in open(p):
        if l.startswith('candidates'): head = l.strip()
        if '|' not in l or l.startswith('taps '): continue
        a, b = l.split('|')
        d[tuple(a.split())] = b.split()
    return head, d
h0, d0 = load('gpurun_out/layers_a.txt'); h1, d1 = load('gpurun_out/layers_b.txt')
print('%s=%s: %s' % (V, A, h0)); print('%s=%s: %s' % (V, B, h1))
print('shape (taps B H W Cin Cout ups pro arb sk)            n   ms/step(%s)  ms/step(%s)' % (A, B))
tot = [0, 0]
for k, v in sorted(d0.items(), key=lambda kv: -float(kv[1][1])):
    if k[0] != T: continue
    a, b = float(v[1]), float(d1.get(k, v)[1])
    tot[0] += a; tot[1] += b
    print(' '.join('%5s' % x for x in k), '%5s' % v[0], '%7.3f %7.3f' % (a, b), '  x%.2f' % (a / b))
print('sum: %.2f -> %.2f ms/step' % tuple(tot))
PY
