"""Would two half-populations on two HIP streams hide each other's latency-bound layers?
One model + loss + optimizer per stream (own plans and arenas), 9 candidates each, the step of each
captured as a HIP graph; per iteration both graphs are launched, one per stream.  Compared with ONE
optimizer stepping 18 candidates and with the two halves one after the other on one stream."""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter('ignore')
from pix2latent_amd import VariableManager, distribution
from pix2latent_amd.utils import synthetic as S, function_hooks as hook
from pix2latent_amd.model.biggan import BigGAN
from pix2latent_amd.optimizer import GradientOptimizer
import pix2latent_amd.loss_functions as LF
dev = 'cuda'
W, Wv = S.biggan_weights(0), S.lpips_vgg_weights(1)
target, weight = S.synthetic_target(256, 1), S.synthetic_weight_mask(256)


def make(n, graph):
    model = BigGAN(weights=W, device=dev)
    loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=Wv, device=dev)
    vm = VariableManager(device=dev)
    vm.register('z', (128,), 'input', distribution=distribution.TruncatedNormalModulo(), learning_rate=0.05,
                hook_fn=hook.Clamp(2.0))
    vm.register('c', (128,), 'input', default=0.05 * torch.randn(128), learning_rate=0.01)
    vm.register('target', (3, 256, 256), 'output', requires_grad=False, default=target)
    vm.register('weight', (3, 256, 256), 'output', requires_grad=False, default=weight)
    opt = GradientOptimizer(model, vm, loss_fn, max_batch_size=9, exec_batch_size=n, use_graph=graph)
    return opt, vm.initialize(num_samples=n)


def timeit(fn, n=10):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


o18, v18 = make(18, False)
o18.step(v18, optimize=True, transform=True)
print('one stream, 18 candidates             : %.2f ms' % timeit(lambda: o18.step(v18, optimize=True)))
s = [torch.cuda.Stream(), torch.cuda.Stream()]
pair = []
for k in range(2):
    with torch.cuda.stream(s[k]):
        o, v = make(9, True)
        o.step(v, optimize=True, transform=True)
        pair.append((o, v))
torch.cuda.synchronize()


def both():
    for k in range(2):
        with torch.cuda.stream(s[k]):
            pair[k][0].step(pair[k][1], optimize=True)


def serial():
    with torch.cuda.stream(s[0]):
        for k in range(2):
            pair[k][0].step(pair[k][1], optimize=True)


print('two streams x 9 candidates (graphs)   : %.2f ms' % timeit(both))
print('one stream, 9 + 9 one after the other : %.2f ms' % timeit(serial))
