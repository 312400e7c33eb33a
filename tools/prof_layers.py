"""Per-launch table of the conv kernel inside one BasinCMA inner step (BigGAN-256, pop 18):
shape, time, algorithmic TFLOP/s and GB/s per launch, grouped by layer shape.
Uses the library's own hipEvent profiler (p2l_prof_dump)."""
import collections
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

DUMP = os.environ.get('P2L_PROF_DUMP', '/tmp/p2l_layers.txt')


def main():
    import time
    import warnings
    warnings.simplefilter('ignore')
    from pix2latent_amd import _native as N
    from pix2latent_amd import VariableManager, distribution
    from pix2latent_amd.utils import synthetic as S, function_hooks as hook
    from pix2latent_amd.model.biggan import BigGAN
    from pix2latent_amd.optimizer import GradientOptimizer
    import pix2latent_amd.loss_functions as LF
    pop = int(os.environ.get('P2L_POP', '18'))     # candidates on this rank (18 | 9 | 5 | 3 | 2)
    dev = torch.device('cuda:0')
    model = BigGAN(weights=S.biggan_weights(0), device=dev)
    loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=S.lpips_vgg_weights(1), device=dev)
    vm = VariableManager(device=dev)
    vm.register('z', (128,), 'input', distribution=distribution.TruncatedNormalModulo(),
                learning_rate=0.05, hook_fn=hook.Clamp(2.0))
    vm.register('c', (128,), 'input', default=0.05 * torch.randn(128), learning_rate=0.01)
    vm.register('target', (3, 256, 256), 'output', requires_grad=False,
                default=S.synthetic_target(256, 1))
    vm.register('weight', (3, 256, 256), 'output', requires_grad=False,
                default=S.synthetic_weight_mask(256))
    opt = GradientOptimizer(model, vm, loss_fn, max_batch_size=9, exec_batch_size=pop)
    variables = vm.initialize(num_samples=pop)
    for i in range(2):
        opt.step(variables, optimize=True, transform=(i == 0))
    torch.cuda.synchronize()
    lib = N.lib()
    N.check(lib.p2l_prof_begin(4096), 'prof_begin')
    N.check(lib.p2l_prof_dump(DUMP.encode()), 'prof_dump')
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.step(variables, optimize=True)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / steps * 1e3
    T = N.prof_end()
    f, m, c, b = T.flops, T.ms, T.count, T.bytes
    lib.p2l_prof_dump(None)
    rows = collections.OrderedDict()
    for line in open(DUMP):
        v = line.split()
        key = tuple(int(x) for x in v[:10]) + (int(v[13]) if len(v) > 13 else 6,)
        fl, by, ms = float(v[10]), float(v[11]), float(v[12])
        r = rows.setdefault(key, [0, 0.0, 0.0, 0.0])
        r[0] += 1; r[1] += fl; r[2] += by; r[3] += ms
    tot = sum(r[3] for r in rows.values())
    print('candidates %d: %.2f ms/step, conv launches %.2f ms/step' % (pop, step_ms, tot / steps))
    print('taps   B    H    W   Cin  Cout ups pro arb sk mm | n/step  ms/step   TFLOP/s    GB/s  share   (mm: 16-bit MFMA products per fp32 product: 6 bf16x3, 3 fp16x2, 16 = fp32 MFMA)')
    for key, r in sorted(rows.items(), key=lambda kv: -kv[1][3]):
        print('%4d %3d %4d %4d %5d %5d %3d %3d %3d %2d %2d | %5.1f %8.3f %9.1f %8.0f %5.1f%%' % (
            key + (r[0] / steps, r[3] / steps, r[1] / r[3] / 1e9, r[2] / r[3] / 1e6, 100 * r[3] / tot)))
    for taps in (9, 1, 4, 16):
        sel = [r for k, r in rows.items() if k[0] == taps]
        if sel:
            ms = sum(r[3] for r in sel)
            print('taps=%d: %.3f ms/step, %.1f TFLOP/s, %.0f GB/s' % (
                taps, ms / steps, sum(r[1] for r in sel) / ms / 1e9, sum(r[2] for r in sel) / ms / 1e6))


if __name__ == '__main__':
    main()
