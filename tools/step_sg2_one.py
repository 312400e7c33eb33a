"""ONE StyleGAN2 configuration of bench.py's config.extra, a few inner steps, with the library's
per-launch conv profiler dumping a per-layer table: the process rocprofv3 wraps for
profiles/round3_sg2_*.   usage: step_sg2_one.py c4 | c5  [layer table path]"""
import collections, ctypes as C, os, sys, time, warnings
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter('ignore')
from pix2latent_amd import VariableManager, distribution, _native as N
from pix2latent_amd.utils import synthetic as S, function_hooks as hook
from pix2latent_amd.model.stylegan2 import StyleGAN2
from pix2latent_amd.optimizer import GradientOptimizer
import pix2latent_amd.loss_functions as LF
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c4'
table = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(2)
loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=S.lpips_vgg_weights(1), device=dev)
vm = VariableManager(device=dev)
if cfg == 'c4':
    gen = StyleGAN2(model='cars', search='z', device=dev)
    fixed = [torch.randn(1, 1, s_[2], s_[3], generator=g).to(dev) for s_ in gen.noise_shape]

    class FixedNoise(torch.nn.Module):
        def forward(self, z=None):
            return gen.forward_z(z, noises=[n_.expand(z.size(0), -1, -1, -1).contiguous() for n_ in fixed])
    model, n, size, kw = FixedNoise(), 32, 512, dict(exec_batch_size='all')
    vm.register('z', (512,), 'input', distribution=distribution.TruncatedNormalModulo(), learning_rate=0.05,
                hook_fn=hook.Compose(hook.NormalPerturb(sigma=0.05), hook.Clamp(2.0)))
    mask = torch.zeros(3, 512, 512); mask[:, 64:-64, :] += 1.0
    for nm, t in (('target', S.synthetic_target(512, 1)), ('weight', torch.ones(3, 512, 512)), ('loss_mask', mask)):
        vm.register(nm, (3, 512, 512), 'output', requires_grad=False, default=t)
else:
    gen = StyleGAN2(model='ffhq', search='w+', device=dev)
    n_noise = sum(s_[-2] * s_[-1] for s_ in gen.noise_shape)
    model, n, size, kw = gen, 3, 1024, dict(use_graph=False)
    vm.register('z', (18, 512), 'input', learning_rate=0.05, default=gen.latent_mean.cpu().view(1, 512).repeat(18, 1))
    vm.register('noises', (n_noise,), 'input', learning_rate=0.05, default=torch.randn(n_noise, generator=g))
    vm.register('target', (3, 1024, 1024), 'output', requires_grad=False, default=S.synthetic_target(1024, 1))
    vm.register('weight', (3, 1024, 1024), 'output', requires_grad=False, default=S.synthetic_weight_mask(1024))
opt = GradientOptimizer(model, vm, loss_fn, max_batch_size=9, **kw)
variables = vm.initialize(num_samples=n)
for i in range(3):
    opt.step(variables, optimize=True, transform=(i == 0))
torch.cuda.synchronize()
lib = N.lib()
dump = '/tmp/p2l_sg2_layers.txt'
N.check(lib.p2l_prof_begin(8192), 'prof_begin')
N.check(lib.p2l_prof_dump(dump.encode()), 'prof_dump')
K = 3
t0 = time.perf_counter()
for _ in range(K):
    opt.step(variables, optimize=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
T = N.prof_end()
f, m, c, b = T.flops, T.ms, T.count, T.bytes
lib.p2l_prof_dump(None)
head = 'StyleGAN2 %s %d^2, %d candidates: %.2f ms/step = %.1f evals/s (with the per-launch profiler)' % (
    'cars' if cfg == 'c4' else 'ffhq', size, n, ms, n / ms * 1e3)
print(head)
rows = collections.OrderedDict()
for line in open(dump):
    v = line.split()
    key = tuple(int(x) for x in v[:10])
    r = rows.setdefault(key, [0, 0.0, 0.0, 0.0])
    r[0] += 1; r[1] += float(v[10]); r[2] += float(v[11]); r[3] += float(v[12])
out = [head, 'conv launches of the step (library hipEvent profiler), %d steps; other kernels: *_kernel_stats.csv' % K,
       'taps   B    H    W   Cin  Cout ups pro arb sk | n/step  ms/step   TFLOP/s    GB/s  share']
tot = sum(r[3] for r in rows.values())
for key, r in sorted(rows.items(), key=lambda kv: -kv[1][3]):
    out.append('%4d %3d %4d %4d %5d %5d %3d %3d %3d %2d | %5.1f %8.3f %9.1f %7.0f %5.1f%%' % (
        key + (r[0] / K, r[3] / K, r[1] / r[3] / 1e9, r[2] / r[3] / 1e6, 100 * r[3] / tot)))
for taps in (9, 1):
    sel = [r for k_, r in rows.items() if k_[0] == taps]
    if sel:
        out.append('taps=%d: %.3f ms/step, %.1f TFLOP/s, %.0f GB/s' % (
            taps, sum(r[3] for r in sel) / K, sum(r[1] for r in sel) / sum(r[3] for r in sel) / 1e9,
            sum(r[2] for r in sel) / sum(r[3] for r in sel) / 1e6))
out.append('conv launches: %.2f ms of the %.2f ms step' % (tot / K, ms))
if table:
    open(table, 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[-4:]))
