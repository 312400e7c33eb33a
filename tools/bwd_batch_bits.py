"""Which gradient tensor depends on the batch composition?  Forward + loss + backward of the bench
problem's first candidates, once inside a batch of 18 and once in smaller batches (9, 5, 3, 2): the
losses, d loss / d image, the per-layer CBN gradients (d s, d t of all 48 CBN layers, read back through
p2l_biggan_ws_lookup) and dz, dc are compared bit for bit; the LAST layer (in forward order) whose d s
/ d t differ names the first kernel of the backward pass that is not batch-independent.

    python tools/bwd_batch_bits.py
"""
import os, sys, contextlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device('cuda:0')
opt, vm, (W, Wv, c_default, target, weight) = bench.build_problem(dev)
with contextlib.redirect_stdout(sys.stderr):
    opt.setup_cma(vm)
    variables = opt.cma_init(vm)
model, loss_fn = opt.model, opt.loss_fn
Z = torch.stack([t.detach() for t in variables.input.z.data]).to(dev).clamp(-2, 2)
Cc = torch.stack([t.detach() for t in variables.input.c.data]).to(dev)
T = target.to(dev).unsqueeze(0)
Wt = weight.to(dev).unsqueeze(0)
d = model._desc
names, offs = [], []
for i in range(d.n_blocks):
    for k in range(4):
        names.append('block%d.bn_%d' % (i, k)); offs.append(d.blocks[i].cbn_off[k])
offs.append(d.cbn_total)


def run(n):
    z = Z[:n].clone().requires_grad_(True)
    c = Cc[:n].clone().requires_grad_(True)
    out = model(z=z, c=c)
    out.retain_grad()
    loss = loss_fn(out, T.expand(n, -1, -1, -1).contiguous(), Wt.expand(n, -1, -1, -1).contiguous())
    (loss / 9.0).sum().backward()
    torch.cuda.synchronize()
    return dict(loss=loss.detach().clone(), dimg=out.grad.clone(), ds=model.saved_activation(4).clone().view(n, -1),
                dt=model.saved_activation(5).clone().view(n, -1), dz=z.grad.clone(), dc=c.grad.clone())


ref = run(18)
for n in [int(v) for v in os.environ.get('BATCHES', '9,5,3,2,1').split(',')]:
    r = run(n)
    line = ['batch %2d vs 18:' % n]
    for key in ('loss', 'dimg', 'dz', 'dc'):
        a, b = ref[key][:n], r[key]
        nd = int((a.view(torch.int32) != b.view(torch.int32)).sum())
        line.append('%s %s' % (key, 'same' if nd == 0 else '%d differ (max |d| %.3g)' % (nd, (a - b).abs().max().item())))
    print('  '.join(line))
    bad = []
    for li, nm in enumerate(names):
        for key in ('ds', 'dt'):
            a, b = ref[key][:n, offs[li]:offs[li + 1]], r[key][:, offs[li]:offs[li + 1]]
            nd = int((a.contiguous().view(torch.int32) != b.contiguous().view(torch.int32)).sum())
            if nd:
                bad.append('%s.%s(%d)' % (nm, key, nd))
    print('    CBN gradients that differ:', ' '.join(bad[-12:]) if bad else 'none', '' if len(bad) <= 12 else '(... %d in all; the last ones in forward order shown)' % len(bad))
