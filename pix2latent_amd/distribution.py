"""Latent initialisation distributions (reference pix2latent/distribution.py)."""
import torch


class TruncatedNormalModulo():
    """
    fmod(N(0, I) + mu, 2.0).

    As in the reference (distribution.py:27-28) the `sigma` and `trunc`
    constructor arguments are accepted but NOT used: sigma is always 1.0 and
    the modulo is always 2.0.
    """

    def __init__(self, mu=0., sigma=1., trunc=2.):
        if type(mu) in [int, float]:
            self.mu = mu
        else:
            self.mu = mu.detach().cpu()
        self.sigma = 1.0
        self.trunc = 2.0
        return

    def __call__(self, num_samples, shape):
        with torch.no_grad():
            _x = self.sigma * torch.randn((num_samples, *shape))
            return torch.fmod(_x + self.mu, self.trunc)


def truncated_clamp_normal(sigma=1.0, trunc=2.0):
    """ N(0, sigma^2) hard-clamped to [-trunc, trunc]
    (the reference version, distribution.py:39-58, raises NameError when called) """
    def _dist_fn(num_samples, shape):
        with torch.no_grad():
            return (sigma * torch.randn((num_samples, *shape))).clamp_(-trunc, trunc)
    return _dist_fn


def normal(sigma=1.0):
    """ N(0, sigma^2) """
    def _dist_fn(num_samples, shape):
        with torch.no_grad():
            return (sigma * torch.randn((num_samples, *shape)))
    return _dist_fn
