"""Which tensor differs first when a repeated forward-only re-score is not bit-reproducible
(VERDICT r4 weak #3): the bench population is re-scored REPS times; after every repetition the
generator arena, the loss arena and the images are compared bit for bit with repetition 0 and the
differing elements are attributed to the named regions of the plans (p2l_biggan_ws_lookup /
p2l_projloss_ws_lookup), in forward order, with the candidates they belong to.

    P2L_LIB_PATH=tools/micro/libp2l_hip_ab.so REPS=60 python tools/ulp_hunt.py
"""
import os, sys, ctypes as C, contextlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pix2latent_amd import _native as N

dev = torch.device('cuda:0')
opt, vm, _ = bench.build_problem(dev)
with contextlib.redirect_stdout(sys.stderr):
    opt.setup_cma(vm)
    variables = opt.cma_init(vm)
for j in range(2):
    opt.step(variables, optimize=True, transform=(j == 0))
model, eng = opt.model, opt.loss_fn._engine
lib = N.lib()


def gen_regions():
    B = model._last_B
    regs = []

    def look(what, L, name):
        off, shape = C.c_size_t(0), (C.c_int32 * 4)()
        if lib.p2l_biggan_ws_lookup(C.byref(model._desc), B, what, L, C.byref(off), shape) != 0:
            return
        regs.append((name, off.value, list(shape)))
    look(2, 0, 'cbn_s'); look(3, 0, 'cbn_t'); look(1, 0, 'gen_z')
    nb = model._desc.n_blocks
    li = 0
    for i in range(nb):
        if i == model._desc.attn_before:
            look(8, 0, 'attn.phi'); look(9, 0, 'attn.g'); look(0, li, 'attn.out(L%d)' % li)
            li += 1
        for kk in range(3):
            look(7, 3 * i + kk, 'block%d.h%d' % (i, kk + 1))
        look(0, li, 'block%d.out(L%d)' % (i, li))
        li += 1
    look(10, 0, 'splitk_ws'); look(11, 0, 'amax_ring')
    return regs


def loss_regions():
    B, H, W = eng.shape
    regs = []
    for idx in range(13):
        off, shape = C.c_size_t(0), (C.c_int32 * 4)()
        if lib.p2l_projloss_ws_lookup(B, H, W, idx, C.byref(off), shape) == 0:
            regs.append(('vgg.conv%d' % idx, off.value, list(shape)))
    return regs


# a re-score runs the reference's chunks (9 + 9): snapshot the arenas after EVERY chunk's forward
cur = {'gen': [], 'loss': [], 'img': []}
_fwd0, _lfwd0 = model._run_forward, eng.f_fwd


def _fwd(z, c):
    out = _fwd0(z, c)
    torch.cuda.synchronize()
    cur['gen'].append(model._ws.view(torch.int32).clone())
    cur['img'].append(model._img16.view(torch.int32).clone())
    return out


def _lfwd(*a):
    rc = _lfwd0(*a)
    torch.cuda.synchronize()
    cur['loss'].append(eng.ws.view(torch.int32).clone())
    return rc


model._run_forward, eng.f_fwd = _fwd, _lfwd


def attribute(name, a, b, regs):
    d = (a != b)
    if not bool(d.any()):
        return []
    out = []
    covered = torch.zeros_like(d)
    for rn, off, shp in regs:
        n = int(np.prod(shp))
        dd = d[off:off + n]
        covered[off:off + n] = True
        cnt = int(dd.sum())
        if cnt:
            per = n // shp[0]
            idx = torch.nonzero(dd).flatten()
            imgs = sorted(set((idx // per).tolist()))
            fa = a[off:off + n].view(torch.float32)[idx[:4]].tolist()
            fb = b[off:off + n].view(torch.float32)[idx[:4]].tolist()
            out.append('%s:%s %s: %d of %d differ, leading index %s: %s; first %s vs %s' %
                       (name, rn, shp, cnt, n, 'set' if rn == 'amax_ring' else 'image', imgs,
                        ['%.9g' % v for v in fa], ['%.9g' % v for v in fb]))
    rest = int((d & ~covered).sum())
    if rest:
        idx = torch.nonzero(d & ~covered).flatten()
        out.append('%s:(unnamed) %d differ, offsets %d .. %d' % (name, rest, int(idx[0]), int(idx[-1])))
    return out


reps = int(os.environ.get('REPS', 40))
losses, first = [], None
nbad = 0
for i in range(reps):
    for v in cur.values():
        del v[:]
    _, l, _ = opt.step(variables, optimize=False)
    losses.append(np.array(l, dtype=np.float64))
    s = {k: list(v) for k, v in cur.items()}
    if first is None:
        first = s
        gr, lr = gen_regions(), loss_regions()
        print('chunks per re-score: %d of %d candidates; generator arena %.2f GB, %d named regions; loss arena %.2f GB' %
              (len(s['gen']), model._last_B, s['gen'][0].numel() * 4e-9, len(gr), s['loss'][0].numel() * 4e-9))
        continue
    rep = []
    for ch in range(len(s['gen'])):
        rep += attribute('chunk%d gen' % ch, first['gen'][ch], s['gen'][ch], gr)
        rep += attribute('chunk%d loss' % ch, first['loss'][ch], s['loss'][ch], lr)
        if bool((first['img'][ch] != s['img'][ch]).any()):
            d = (first['img'][ch] != s['img'][ch]).view(first['img'][ch].shape[0] if first['img'][ch].dim() > 1 else -1, -1)
            rep.append('chunk%d img16 differs (%d elements)' % (ch, int(d.sum())))
    if rep:
        nbad += 1
        if nbad <= 6:
            print('--- repetition %d differs from repetition 0:' % i)
            for r in rep:
                print('   ', r)
res = np.stack(losses)
print(os.environ.get('P2L_LIB_PATH', 'product'), 'repetitions that differ:', nbad, 'of', reps - 1,
      '| distinct loss vectors:', len(np.unique(res, axis=0)),
      '| candidates that ever differ:', np.nonzero((res != res[0]).any(axis=0))[0].tolist())
