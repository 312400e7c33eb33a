"""In-place latent hooks (API of reference pix2latent/utils/function_hooks.py:10-126:
`Clamp`, `Normalize`, `NormalPerturb`, `ScheduledNormalPerturb`, `Compose`; a hook is
called as `hook(list_of_per_sample_tensors)` before every forward pass and mutates the
tensors in place).

Device form.  The per-sample tensors of a variable are views of one contiguous
[N, *shape] buffer, so a hook also offers `apply_batched(rows, span)`: one kernel over the
rows instead of one per sample.

One `apply_batched` call stands for ONE hook call of the reference, i.e. one chunk of
`max_batch_size` samples.  `span` (a `HookSpan`) names that reference chunk [c0, c1) and
which of its rows [start, stop) are present.  This makes the random hooks independent of how
the population is executed and partitioned:

  * the reference draws `randn_like(sample)` sample by sample from the default generator;
    the row form draws exactly that sequence, one draw per row of the chunk in order;
  * rows of the chunk that are NOT present (they belong to another rank of a sharded run)
    are drawn and discarded, and a rank owning nothing of a chunk discards all of it.  Every
    rank therefore consumes the generator exactly like a single process: equally seeded, a
    run on any number of GPUs perturbs candidate i with the same numbers
    (tests/test_parallel_gloo.py);
  * `ScheduledNormalPerturb`'s clock ticks once per call = once per reference chunk.

The closure applies the hooks of a step chunk by chunk over the whole population before the
first forward pass of the step (optimizer/closure.py `apply_hooks`).
"""
import math
from collections import namedtuple

import torch

# rows [start, stop) of the reference chunk [c0, c1) (population row numbers)
HookSpan = namedtuple('HookSpan', 'start stop c0 c1')


def whole(rows):
    """span of a buffer that is one complete reference chunk"""
    return HookSpan(0, rows.size(0), 0, rows.size(0))


class _Hook(object):
    """per-sample call form derived from the row form"""

    #: True when the hook consumes random numbers
    stochastic = False
    #: True when a captured HIP graph of the hook may be replayed verbatim (nothing about its
    #: launches depends on host state that changes from step to step)
    graph_safe = True

    def __call__(self, vars):
        for v in vars:
            self.on_rows(v.data.unsqueeze(0))
        self.end_of_call()
        return

    def apply_batched(self, rows, span=None):
        span = span or whole(rows)
        if self.stochastic:
            self.discard(rows, span.start - span.c0)
        if span.stop > span.start:
            self.on_rows(rows)
        if self.stochastic:
            self.discard(rows, span.c1 - span.stop)
        self.end_of_call()

    # -- overridables ------------------------------------------------------------------
    def on_rows(self, rows):
        raise NotImplementedError

    def discard(self, rows, count):
        pass

    def end_of_call(self):
        pass


class Clamp(_Hook):
    """ clamps the variable to [-trunc, trunc] """

    def __init__(self, trunc):
        self.trunc = trunc

    def on_rows(self, rows):
        rows.clamp_(-self.trunc, self.trunc)


class Normalize(_Hook):
    """ standardises each sample to zero mean / unit (unbiased) deviation -- the latent
    normalisation of StyleGAN2.  `mu` / `std` are accepted and ignored like the reference's """

    def __init__(self, mu=0., std=1.):
        self.mu, self.std = mu, std

    def on_rows(self, rows):
        flat = rows.reshape(rows.size(0), -1)
        mean, std = flat.mean(1, keepdim=True), flat.std(1, keepdim=True)
        flat.sub_(mean).div_(std)


class NormalPerturb(_Hook):
    """ adds N(0, sigma^2) noise """

    stochastic = True

    def __init__(self, sigma=0.1):
        self.sigma = sigma

    def strength(self):
        return self.sigma

    def on_rows(self, rows):
        # one draw per row, in row order: the reference's generator consumption
        noise = torch.stack([torch.randn_like(rows[i]) for i in range(rows.size(0))])
        rows.add_(self.strength() * noise)

    def discard(self, rows, count):
        like = rows.new_empty(rows.shape[1:])
        for _ in range(count):
            torch.randn_like(like)


class ScheduledNormalPerturb(NormalPerturb):
    """ noise whose strength decays to zero over `max_step` hook calls:
    (sigma * max(0, 1 - t / (max_step - 1))) ** 2, t = number of calls so far.  (The
    reference class, function_hooks.py:73-102, fails on its un-imported `math`; like it,
    `pow` is accepted but the exponent is always 2.) """

    graph_safe = False           # the strength is a host number that changes every call

    def __init__(self, sigma=0.1, max_step=500, pow=2):
        NormalPerturb.__init__(self, sigma)
        self.max_step = max_step
        self.t = 0
        self.pow = 2

    def strength(self):
        progress = self.t / (float(self.max_step) - 1)
        return math.pow(self.sigma * max(0, 1 - progress), self.pow)

    def end_of_call(self):
        self.t += 1


class Compose(_Hook):
    """ applies hooks one after the other """

    def __init__(self, *hook_fns):
        self.hook_fns = hook_fns

    @property
    def graph_safe(self):
        return all(getattr(fn, 'graph_safe', False) for fn in self.hook_fns)

    def __call__(self, vars):
        for fn in self.hook_fns:
            fn(vars)
        return

    def apply_batched(self, rows, span=None):
        span = span or whole(rows)
        for fn in self.hook_fns:
            if hasattr(fn, 'apply_batched'):
                fn.apply_batched(rows, span)
            elif span.stop > span.start:
                fn(list(rows))
