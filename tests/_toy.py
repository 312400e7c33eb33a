"""Tiny deterministic differentiable 'generator' used to pin the orchestration
layer (closure / optimizers / VariableManager) against golden traces captured
from the imported reference (tools/make_golden.py)."""
import types

import numpy as np
import torch
import torch.nn as nn


class ToyGenerator(nn.Module):
    """z[B,6], c[B,4] -> [B,3,4,4] in (-1,1)"""

    def __init__(self, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.Wz = nn.Parameter(torch.randn(6, 48, generator=g) * 0.4)
        self.Wc = nn.Parameter(torch.randn(4, 48, generator=g) * 0.4)
        self.calls = []

    def forward(self, z=None, c=None):
        self.calls.append((int(z.size(0)), bool(torch.is_grad_enabled())))
        return torch.tanh(z @ self.Wz + c @ self.Wc).view(-1, 3, 4, 4)


def toy_target(seed=3):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(3, 4, 4, generator=g) * 2 - 1


def toy_weight():
    w = torch.ones(3, 4, 4)
    w[:, :2] = 0.3
    return w


class FakeCMAES(object):
    """recording stand-in for cma.CMAEvolutionStrategy: popsize 4+floor(3 ln N),
    ask() = mean + sigma * seeded normal, tell() records and moves the mean to the
    best sample (enough to pin WHAT the reference feeds to tell)."""
    log = []

    def __init__(self, x0, sigma0, opts=None):
        self.N = len(x0)
        self.mean = np.array(x0, dtype=np.float64)
        self.sigma = sigma0
        self.sp = types.SimpleNamespace(popsize=4 + int(3 * np.log(self.N)))
        self.rng = np.random.RandomState(1234)
        self.opts = opts or {}

    def ask(self, number=None):
        n = number or self.sp.popsize
        return [self.mean + self.sigma * self.rng.randn(self.N) for _ in range(n)]

    def tell(self, x, y):
        FakeCMAES.log.append((np.array(x).copy(), np.array(y, dtype=np.float64).copy()))
        self.mean = np.array(x[int(np.argmin(y))], dtype=np.float64)




class FakeNGOpt(object):
    """recording stand-in for a nevergrad optimizer (API used by the reference,
    base_ng_optimizer.py:81-83,107,169): ask() -> candidate whose .args is the 1-tuple
    (array of init's shape,), tell(candidate, loss).  ask = mean + seeded normal; tell records
    and moves the mean to the best candidate told so far."""
    log = []          # (kind, payload) in call order, shared across instances
    instances = []

    class Cand(object):
        def __init__(self, value, uid):
            self.args = (value,)
            self.kwargs = {}
            self.uid = uid

    def __init__(self, parametrization=None, budget=None, num_workers=1, **kw):
        self.init = np.array(parametrization.init, dtype=np.float64)
        self.budget = budget
        self.mean = self.init.copy()
        self.best = np.inf
        self.rng = np.random.RandomState(4321)
        self.n_ask = 0
        FakeNGOpt.instances.append(self)

    def ask(self):
        self.n_ask += 1
        c = FakeNGOpt.Cand(self.mean + self.rng.randn(*self.init.shape), self.n_ask)
        FakeNGOpt.log.append(('ask', c.uid))
        return c

    def tell(self, cand, loss):
        loss = float(loss)
        FakeNGOpt.log.append(('tell', (cand.uid, np.array(cand.args[0]).copy(), loss)))
        if loss < self.best:
            self.best, self.mean = loss, np.array(cand.args[0], dtype=np.float64)


class _FakeNGArray(object):
    def __init__(self, init=None, shape=None):
        self.init = np.zeros(shape) if init is None else np.array(init, dtype=np.float64)

    def set_mutation(self, sigma=1.0):
        return self


def fake_nevergrad():
    """module-like object with the attributes the reference touches"""
    ng = types.SimpleNamespace()
    ng.optimizers = types.SimpleNamespace(registry={'CMA': FakeNGOpt, 'DE': FakeNGOpt})
    ng.p = types.SimpleNamespace(Array=_FakeNGArray)
    return ng
