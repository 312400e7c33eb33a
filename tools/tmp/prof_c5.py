import os, sys, warnings, torch
sys.path.insert(0, '.')
warnings.simplefilter('ignore')
from pix2latent_amd import VariableManager
from pix2latent_amd.utils import synthetic as S
from pix2latent_amd.model.stylegan2 import StyleGAN2
from pix2latent_amd.optimizer import GradientOptimizer
import pix2latent_amd.loss_functions as LF
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(2)
loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=S.lpips_vgg_weights(1), device=dev)
vm = VariableManager(device=dev)
gen = StyleGAN2(model='ffhq', search='w+', device=dev)
n_noise = sum(s_[-2] * s_[-1] for s_ in gen.noise_shape)
vm.register('z', (18, 512), 'input', learning_rate=0.05, default=gen.latent_mean.cpu().view(1, 512).repeat(18, 1))
vm.register('noises', (n_noise,), 'input', learning_rate=0.05, default=torch.randn(n_noise, generator=g))
vm.register('target', (3, 1024, 1024), 'output', requires_grad=False, default=S.synthetic_target(1024, 1))
vm.register('weight', (3, 1024, 1024), 'output', requires_grad=False, default=S.synthetic_weight_mask(1024))
opt = GradientOptimizer(gen, vm, loss_fn, max_batch_size=9, use_graph=False)
variables = vm.initialize(num_samples=3)
for i in range(3):
    opt.step(variables, optimize=True, transform=(i == 0))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    opt.step(variables, optimize=True)
    torch.cuda.synchronize()
for e in prof.key_averages(group_by_input_shape=True):
    if any(k in e.key for k in ('copy', 'Memcpy', 'clone', 'cat', 'contiguous', 'aten::to', 'memcpy')):
        print('%-40s calls %3d  cuda %8.1f us  shapes %s' % (e.key[:40], e.count, e.device_time_total if hasattr(e, 'device_time_total') else e.cuda_time_total, str(e.input_shapes)[:150]))
import torch.autograd.profiler as ap
for e in prof.events():
    if 'Memcpy' in e.name or 'copyBuffer' in e.name:
        print('EV', e.name, e.device_time if hasattr(e, 'device_time') else e.cuda_time)

