"""whole inner step (generator + ProjectionLoss + backward + Adam) for the StyleGAN2
configurations: C4 = cars 512^2, 32 samples in chunks of 9; C5 shard = ffhq 1024^2, 3 per GPU"""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter('ignore')
from pix2latent_amd import VariableManager, distribution
from pix2latent_amd.utils import synthetic as S, function_hooks as hook
from pix2latent_amd.model.stylegan2 import StyleGAN2
from pix2latent_amd.optimizer import GradientOptimizer
import pix2latent_amd.loss_functions as LF
dev = 'cuda'
for name, size, n, mb in (('cars', 512, 32, 9), ('cars', 512, 9, 9), ('ffhq', 1024, 3, 3)):
    model = StyleGAN2(model=name, search='z', device=dev)
    loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=S.lpips_vgg_weights(1), device=dev)
    target = S.synthetic_target(size, 1)
    weight = torch.ones(3, size, size)
    loss_mask = torch.zeros(3, size, size); loss_mask[:, size // 8:-size // 8, :] += 1.0
    vm = VariableManager(device=dev)
    vm.register('z', (512,), 'input', distribution=distribution.TruncatedNormalModulo(1.0, 2.0), learning_rate=0.05,
                hook_fn=hook.Compose(hook.NormalPerturb(sigma=0.05), hook.Clamp(2.0)), grad_free=True)
    for nm, t in (('target', target), ('weight', weight), ('loss_mask', loss_mask)):
        vm.register(nm, (3, size, size), 'output', requires_grad=False, default=t)
    ebs = os.environ.get('P2L_EXEC') or None        # candidates per device pass: int | 'all'
    ebs = int(ebs) if ebs and ebs != 'all' else ebs
    opt = GradientOptimizer(model, vm, loss_fn, max_batch_size=mb, exec_batch_size=ebs)
    variables = vm.initialize(num_samples=n)
    for i in range(2):
        opt.step(variables, optimize=True, transform=(i == 0))
    torch.cuda.synchronize(); t = time.perf_counter()
    K = 4
    for _ in range(K):
        opt.step(variables, optimize=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / K * 1e3
    print('StyleGAN2-%s %d^2, %d samples (chunks of %d): %.1f ms/step  %.1f evals/s' % (name, size, n, mb, ms, n / ms * 1e3))
    del model, loss_fn, opt, variables, vm
    torch.cuda.empty_cache()
