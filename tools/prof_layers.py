"""Per-launch table of the conv kernel inside one BasinCMA inner step (BigGAN-256, pop 18):
shape, time, algorithmic TFLOP/s and GB/s per launch, grouped by layer shape.
Uses the library's own hipEvent profiler (P2L_PROF_DUMP)."""
import collections
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

DUMP = os.environ.setdefault('P2L_PROF_DUMP', '/tmp/p2l_layers.txt')


def main():
    import bench
    from pix2latent_amd import _native as N
    pop = 18
    dev = torch.device('cuda:0')
    assert pop == bench.POP
    torch.manual_seed(0)
    opt, vm, _ = bench.build_problem(dev, exec_batch_size=pop)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        opt.setup_cma(vm)
        variables = opt.cma_init(vm)
    for i in range(2):
        opt.step(variables, optimize=True, transform=(i == 0))
    torch.cuda.synchronize()
    lib = N.lib()
    N.check(lib.p2l_prof_begin(4096), 'prof_begin')
    steps = 3
    for _ in range(steps):
        opt.step(variables, optimize=True)
    torch.cuda.synchronize()
    f, m, c, b = (C.c_double * 2)(), (C.c_double * 2)(), (C.c_int32 * 2)(), (C.c_double * 2)()
    N.check(lib.p2l_prof_end2(f, m, c, b), 'prof_end2')
    rows = collections.OrderedDict()
    for line in open(DUMP):
        v = line.split()
        key = tuple(int(x) for x in v[:10])
        fl, by, ms = float(v[10]), float(v[11]), float(v[12])
        r = rows.setdefault(key, [0, 0.0, 0.0, 0.0])
        r[0] += 1; r[1] += fl; r[2] += by; r[3] += ms
    tot = sum(r[3] for r in rows.values())
    print('taps   B    H    W   Cin  Cout ups pro arb sk | n/step  ms/step   TFLOP/s    GB/s  share')
    for key, r in sorted(rows.items(), key=lambda kv: -kv[1][3]):
        print('%4d %3d %4d %4d %5d %5d %3d %3d %3d %2d | %5.1f %8.3f %9.1f %8.0f %5.1f%%' % (
            key + (r[0] / steps, r[3] / steps, r[1] / r[3] / 1e9, r[2] / r[3] / 1e6, 100 * r[3] / tot)))
    for taps in (9, 1, 4, 16):
        sel = [r for k, r in rows.items() if k[0] == taps]
        if sel:
            ms = sum(r[3] for r in sel)
            print('taps=%d: %.3f ms/step, %.1f TFLOP/s, %.0f GB/s' % (
                taps, ms / steps, sum(r[1] for r in sel) / ms / 1e9, sum(r[2] for r in sel) / ms / 1e6))


if __name__ == '__main__':
    main()
