"""In-place latent hooks (reference pix2latent/utils/function_hooks.py:10-126).

A hook is called as `hook(list_of_per_sample_tensors)` and mutates them in
place, exactly like the reference.  Each class additionally offers
`apply_batched(buf)` on the contiguous [N, *shape] buffer that backs those
per-sample tensors, which the closure uses to run one kernel for the whole
chunk instead of N tiny ones.
"""
import math

import torch


class Clamp():
    """ clamps the variable by the specified truncation value """

    def __init__(self, trunc):
        self.trunc = trunc
        return

    def __call__(self, vars):
        for v in vars:
            v.data.clamp_(-self.trunc, self.trunc)
        return

    def apply_batched(self, buf):
        buf.clamp_(-self.trunc, self.trunc)


class Normalize():
    """ standardises each sample to mean 0 / std 1 (StyleGAN2 latent normalisation) """

    def __init__(self, mu=0., std=1.):
        self.mu = mu
        self.std = std
        return

    def __call__(self, vars):
        for v in vars:
            mean = v.mean()
            std = v.std()
            v.data.add_(-mean).div_(std)
        return

    def apply_batched(self, buf):
        flat = buf.view(buf.size(0), -1)
        mean = flat.mean(1, keepdim=True)
        std = flat.std(1, keepdim=True)
        flat.sub_(mean).div_(std)


class NormalPerturb():
    """ perturbs the data with N(0, sigma^2) noise """

    def __init__(self, sigma=0.1):
        self.sigma = sigma
        return

    def __call__(self, vars):
        for v in vars:
            v.data.add_(self.sigma * torch.randn_like(v))
        return

    def apply_batched(self, buf):
        buf.add_(self.sigma * torch.randn_like(buf))


class ScheduledNormalPerturb():
    """ noise decaying from sigma to 0 over max_step calls
    (reference function_hooks.py:73-102; its un-imported `math` is imported here) """

    def __init__(self, sigma=0.1, max_step=500, pow=2):
        self.sigma = sigma
        self.max_step = max_step
        self.t = 0
        self.pow = 2
        return

    def _strength(self):
        p = self.t / (float(self.max_step) - 1)
        return math.pow(self.sigma * max(0, 1 - p), self.pow)

    def __call__(self, vars):
        for v in vars:
            v.data.add_(self._strength() * torch.randn_like(v))
        self.t += 1
        return

    def apply_batched(self, buf):
        buf.add_(self._strength() * torch.randn_like(buf))
        self.t += 1


class Compose():
    """ applies hooks sequentially """

    def __init__(self, *hook_fns):
        self.hook_fns = hook_fns
        return

    def __call__(self, vars):
        for fn in self.hook_fns:
            fn(vars)
        return

    def apply_batched(self, buf):
        for fn in self.hook_fns:
            if hasattr(fn, 'apply_batched'):
                fn.apply_batched(buf)
            else:
                fn(list(buf))
