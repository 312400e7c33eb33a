"""ORACLE (test infrastructure): record / replay of the DISCRETE decisions of a forward pass.

The gradients of this pipeline are piecewise: which side of zero every ReLU input sits on, and
which element of every 2x2 window wins its max-pool, are decisions that two arithmetics (fp32 vs
fp64, torch vs the HIP kernels) take differently for a handful of the ~10^8 units, and every flip
changes the gradient by a step.  To compare ARITHMETIC one has to hold the decisions fixed: a
`DecisionTape` in replay mode makes `relu` multiply by a given 0/1 mask and `maxpool2` pick a given
winner, so that two runs with the same tape differ by rounding only
(tests/test_fixed_mask_grad_gpu.py: the tape is read back from the native run).

Only tests/ may import this module."""
import torch
import torch.nn.functional as F


class DecisionTape(object):
    def __init__(self, replay=None):
        self.items = [] if replay is None else list(replay)
        self.replaying = replay is not None
        self.pos = 0

    def _next(self, name, kind):
        n, k, m = self.items[self.pos]
        assert (n, k) == (name, kind), 'tape out of step: wanted %s/%s, holds %s/%s' % (name, kind, n, k)
        self.pos += 1
        return m

    def relu(self, x, name):
        if not self.replaying:
            self.items.append((name, 'relu', (x > 0).detach()))
            return F.relu(x)
        return x * self._next(name, 'relu').to(x.dtype)

    def abs_diff(self, out, target, name):
        """|target - out|: the decision is the sign of (out - target) per element"""
        if not self.replaying:
            self.items.append((name, 'sign', torch.sign(out - target).detach()))
            return torch.abs(target - out)
        return (out - target) * self._next(name, 'sign').to(out.dtype)

    def maxpool2(self, x, name):
        """2x2 / stride 2; the decision is a one-hot mask over every window (same shape as x)"""
        if not self.replaying:
            y, idx = F.max_pool2d(x, 2, 2, return_indices=True)
            self.items.append((name, 'pool', winner_mask(x, idx)))
            return y
        m = self._next(name, 'pool').to(x.dtype)
        return F.avg_pool2d(x * m, 2, 2) * 4.0           # the one live element of each window

    def maxpool3s2(self, x, name):
        """3x3 / stride 2, ceil_mode (lpips SqueezeNet); replay gathers the recorded winner of every window"""
        if not self.replaying:
            y, idx = F.max_pool2d(x, 3, 2, ceil_mode=True, return_indices=True)
            self.items.append((name, 'pool3', idx.detach()))
            return y
        idx = self._next(name, 'pool3')
        return x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)


def pool3s2_winners(x):
    """flat indices [B,C,Ho,Wo] of the winners of a 3x3 / stride-2 max-pool over whole windows (torch's
    first-maximum tie rule): the windows overlap, so the decision is one index per WINDOW"""
    _, idx = F.max_pool2d(x, 3, 2, return_indices=True)
    return idx


def winner_mask(x, idx=None):
    """one-hot mask [B,C,H,W] of the max-pool winners of x (torch's first-maximum tie rule)"""
    if idx is None:
        _, idx = F.max_pool2d(x, 2, 2, return_indices=True)
    m = torch.zeros(x.shape[0], x.shape[1], x.shape[2] * x.shape[3], dtype=torch.bool, device=x.device)
    m.scatter_(2, idx.flatten(2), True)
    return m.view_as(x)


def relu(x, tape, name):
    return F.relu(x) if tape is None else tape.relu(x, name)


def maxpool2(x, tape, name):
    return F.max_pool2d(x, 2, 2) if tape is None else tape.maxpool2(x, name)
