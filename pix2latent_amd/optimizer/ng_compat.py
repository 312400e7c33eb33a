"""Minimal ask/tell façade with the shape of the `nevergrad` API the reference
uses (pix2latent/optimizer/base_ng_optimizer.py:1,33,81-83,107,169):

    opt = registry[method](parametrization=p.Array(init=mu), budget=n)
    cand = opt.ask();  cand.args -> (ndarray,) ;  opt.tell(cand, loss)

nevergrad>=0.4.0.post3 (requirements.txt:3) is absent here; PARITY UNPINNED.
Methods: 'CMA' (our CMA-ES, tells buffered per generation like nevergrad's
wrapper around pycma, arbitrary number of asks in flight), 'RandomSearch',
'OnePlusOne', 'DE'/'TwoPointsDE'.
"""
import math

import numpy as np

from .cma_es import CMAEvolutionStrategy


class Array(object):
    def __init__(self, init=None, shape=None):
        self.init = np.zeros(shape) if init is None else np.array(init, dtype=np.float64)
        self.sigma = 1.0

    def set_mutation(self, sigma=1.0):
        self.sigma = float(sigma)
        return self


class Candidate(object):
    """nevergrad's candidate for a p.Array parametrisation: `args` is a 1-tuple holding the
    array in the shape of `init` (the reference stacks them with
    np.concatenate([x.args for x in asked]), base_ng_optimizer.py:107)."""

    def __init__(self, value, uid):
        self.value = value
        self.args = (value,)
        self.kwargs = {}
        self.uid = uid


class _Base(object):
    def __init__(self, parametrization, budget=None, num_workers=1, seed=None):
        self.p = parametrization
        self.budget = budget
        self.dim = self.p.init.size
        self.rng = np.random.RandomState(seed)
        self._uid = 0
        self.num_ask = 0
        self.num_tell = 0
        self.best = (np.inf, None)

    def _new(self, z):
        self._uid += 1
        self.num_ask += 1
        c = Candidate((self.p.init.reshape(-1) + self.p.sigma * z).reshape(self.p.init.shape),
                      self._uid)
        c._z = z
        return c

    def _note(self, cand, loss):
        self.num_tell += 1
        if loss < self.best[0]:
            self.best = (float(loss), cand)

    def provide_recommendation(self):
        return self.best[1]


class RandomSearch(_Base):
    def ask(self):
        return self._new(self.rng.randn(self.dim))

    def tell(self, cand, loss):
        self._note(cand, float(loss))


class OnePlusOne(_Base):
    def __init__(self, *a, **k):
        super(OnePlusOne, self).__init__(*a, **k)
        self.parent = np.zeros(self.dim)
        self.parent_loss = np.inf
        self.step = 1.0

    def ask(self):
        if self.num_ask == 0:
            return self._new(self.parent.copy())
        return self._new(self.parent + self.step * self.rng.randn(self.dim))

    def tell(self, cand, loss):
        loss = float(loss)
        self._note(cand, loss)
        if loss <= self.parent_loss:
            self.parent, self.parent_loss = cand._z.copy(), loss
            self.step *= 2.0
        else:
            self.step *= 0.84


class DE(_Base):
    def __init__(self, *a, **k):
        super(DE, self).__init__(*a, **k)
        self.popsize = max(30, 4 + int(3 * math.log(self.dim)))
        self.pop = []     # (z, loss)

    def ask(self):
        if len(self.pop) < self.popsize:
            return self._new(self.rng.randn(self.dim))
        idx = self.rng.choice(len(self.pop), 3, replace=False)
        a, b, c = (self.pop[i][0] for i in idx)
        donor = a + 0.8 * (b - c)
        target = self.pop[self.rng.randint(len(self.pop))][0]
        cross = self.rng.rand(self.dim) < 0.5
        cand = self._new(np.where(cross, donor, target))
        return cand

    def tell(self, cand, loss):
        loss = float(loss)
        self._note(cand, loss)
        if len(self.pop) < self.popsize:
            self.pop.append((cand._z, loss))
            return
        worst = int(np.argmax([l for _, l in self.pop]))
        if loss < self.pop[worst][1]:
            self.pop[worst] = (cand._z, loss)


class CMA(_Base):
    """generation-buffered CMA-ES: asks are drawn from the current generation's
    sample list (refilled when exhausted), tells are buffered until `popsize`
    of them are available and then passed to the strategy at once."""

    def __init__(self, *a, **k):
        super(CMA, self).__init__(*a, **k)
        self.es = CMAEvolutionStrategy(np.zeros(self.dim), 1.0,
                                       {'seed': self.rng.randint(2 ** 31 - 1)})
        self.popsize = self.es.sp.popsize
        self._to_ask = []
        self._told = []

    def ask(self):
        if not self._to_ask:
            self._to_ask = list(self.es.ask())
        return self._new(self._to_ask.pop(0))

    def tell(self, cand, loss):
        loss = float(loss)
        self._note(cand, loss)
        self._told.append((cand._z, loss))
        if len(self._told) >= self.popsize:
            xs = [z for z, _ in self._told]
            ls = [l for _, l in self._told]
            self.es.tell(xs, ls)
            self._told = []
            self._to_ask = []


registry = {'CMA': CMA, 'RandomSearch': RandomSearch, 'OnePlusOne': OnePlusOne,
            'DE': DE, 'TwoPointsDE': DE}


class _P(object):
    Array = Array


class _Optimizers(object):
    pass


p = _P()
optimizers = _Optimizers()
optimizers.registry = registry
