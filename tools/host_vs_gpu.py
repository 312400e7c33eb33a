"""is the inner step host-bound at small local batch? compare CPU enqueue time with total time"""
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
model = BigGAN(weights=W, device=dev)
loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=Wv, device=dev)
target, weight = S.synthetic_target(256, 1), S.synthetic_weight_mask(256)
for n in (18, 3, 2):
    vm = VariableManager(device=dev)
    vm.register('z', (128,), 'input', distribution=distribution.TruncatedNormalModulo(), learning_rate=0.05, hook_fn=hook.Clamp(2.0))
    vm.register('c', (128,), 'input', default=0.05 * torch.randn(128), learning_rate=0.01)
    vm.register('target', (3, 256, 256), 'output', requires_grad=False, default=target)
    vm.register('weight', (3, 256, 256), 'output', requires_grad=False, default=weight)
    opt = GradientOptimizer(model, vm, loss_fn, max_batch_size=9, exec_batch_size=n)
    variables = vm.initialize(num_samples=n)
    for i in range(3):
        opt.step(variables, optimize=True, transform=(i == 0))
    torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter()
    for _ in range(K):
        opt.step(variables, optimize=True)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print('n=%2d: enqueue %.2f ms/step, total %.2f ms/step -> %s' % (n, t_enq / K * 1e3, t_all / K * 1e3,
          'HOST-bound' if t_enq > 0.9 * t_all else 'GPU-bound (queue runs ahead)'))
