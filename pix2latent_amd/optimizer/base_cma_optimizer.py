"""pycma-style generation sampler for CMAOptimizer / BasinCMAOptimizer /
TransformBasinCMAOptimizer.

API kept from reference pix2latent/optimizer/base_cma_optimizer.py: the `CMA` facade
(`batch_size / ask / tell / mean`, :145-215) and the mixin entry points
`setup_cma / cma_init / cma_update` with attributes `num_samples`, `cma_optimizers`
(:28-141).  The implementation is a `PopulationSampler` strategy (search_loop.py): it owns
ask, the broadcast of rank 0's ask to the other ranks, and the Baldwinian tell.  The
evolution strategy is the installed `cma` package when it imports -- what the reference
instantiates, base_cma_optimizer.py:2,176 -- and optimizer/cma_es.py otherwise
(optimizer/backends.py; PARITY UNPINNED for the fallback).
"""
import numpy as np
import torch

from . import backends
from .search_loop import PopulationSampler, find_grad_free
from ..utils.misc import cprint

CMAEvolutionStrategy, CMA_BACKEND = backends.cma_strategy()
CMA_EXTERNAL = not CMA_BACKEND.startswith('in-tree')


class CMA(object):
    """Facade over the evolution strategy with the reference's four methods.

    A one-dimensional problem is embedded in two dimensions (the strategy needs N >= 2):
    the mean is duplicated, covariance adaptation is switched off (`CMA_on = 0`) and only
    column 0 is shown to the caller, who must hand the same proxy array back to `tell`."""

    def __init__(self, mu=128 * [0], sigma=1.0, seed=None):
        x0 = np.atleast_1d(np.asarray(mu, dtype=np.float64)).ravel()
        opts = {} if seed is None else {'seed': seed}
        self.is_scalar = (x0.size == 1)
        if self.is_scalar:
            x0 = np.repeat(x0, 2)
            opts['CMA_on'] = 0
        self.cma = CMAEvolutionStrategy(x0, sigma, opts)
        self._full = self._proxy = None

    def batch_size(self):
        return self.cma.sp.popsize

    def ask(self, batch_size=None):
        drawn = np.array(self.cma.ask(batch_size))
        if not self.is_scalar:
            return drawn
        self._full, self._proxy = drawn, drawn[:, :1]
        return self._proxy

    def tell(self, x, y):
        if self.is_scalar:
            assert x is self._proxy or np.array_equal(x, self._proxy), \
                'tell() must receive the array ask() returned'
            x = self._full
        return self.cma.tell(x, y)

    def mean(self):
        m = self.cma.mean
        return m[:1] if self.is_scalar else m


class PycmaSampler(PopulationSampler):
    """fixed-population generation sampler: one ask() per generation, one tell()"""

    def __init__(self, var_type, var_name, mu, sigma, seed=None):
        PopulationSampler.__init__(self, var_type, var_name)
        self.shape = tuple(np.shape(mu))
        self.es = CMA(np.asarray(mu, dtype=np.float64).reshape(-1), sigma=sigma, seed=seed)
        self.population = self.es.batch_size()
        self._replica = False

    def _ask(self, n):
        assert n == self.population, 'PyCMA optimizer has fixed sample size'
        asked = self.es.ask()
        return np.asarray(asked).reshape((n,) + self.shape), asked

    def draw(self, variables, shard=None):
        self._replica = bool(CMA_EXTERNAL and shard is not None and shard.enabled and shard.rank != 0)
        values = PopulationSampler.draw(self, variables, shard)      # (a replica is not asked either)
        if self._replica:
            return values
        if shard is not None and shard.enabled and not CMA_EXTERNAL:
            # every replica tells rank 0's population, so the replicas stay identical.  (An EXTERNAL
            # strategy has no told replicas: rank 0 keeps the handle and the 2-d proxy of its own ask,
            # and must NOT enter a collective the replicas -- returned above -- never join.)
            self._handle = values.reshape(len(values), -1)
            if self.es.is_scalar:
                # a scalar variable lives in a 2-d proxy problem whose second coordinate the
                # strategy draws on its own: step-size adaptation sees BOTH coordinates, so every
                # replica must tell rank 0's full draw, not a copy of column 0 (the trajectory
                # must not depend on the number of GPUs)
                self.es._proxy = self._handle
                self.es._full = shard.broadcast_numpy(np.asarray(self.es._full, dtype=np.float64), src=0)
        return values

    def _tell(self, asked, losses):
        if self._replica:
            # an INSTALLED pycma looks told solutions up in its archive of sent ones; a replica
            # never sent rank 0's population, would take the geno / repair path and could drift
            # away from rank 0.  Only rank 0's strategy is read (draw() broadcasts its ask), so
            # the replicas of an external backend are simply not told.  The in-tree strategy
            # (cma_es.py) is a pure function of what it is told and stays replicated: the gloo
            # tests compare its traces on 1 / 2 / 3 ranks.
            return
        self.es.tell(asked, losses)


class _BaseCMAOptimizer(object):
    """mixin used together with _BaseOptimizer"""

    def __init__(self):
        self.num_samples = -1
        self.cma_optimizers = {}      # {(var_type, name): PycmaSampler}
        self.cma_seed = None

    @property
    def sampler(self):
        return next(iter(self.cma_optimizers.values()))

    @torch.no_grad()
    def setup_cma(self, var_manager):
        """one sampler per variable registered with `grad_free`; exactly one is supported"""
        for var_type, name, mu, sigma in find_grad_free(var_manager):
            s = PycmaSampler(var_type, name, mu, sigma, seed=self.cma_seed)
            self.cma_optimizers[(var_type, name)] = s
            self.num_samples = max(self.num_samples, s.population)

        cprint('(cma-es) number of samples: {}  [{}]'.format(self.num_samples, CMA_BACKEND), 'y')

        assert len(self.cma_optimizers.keys()) == 1, \
            'currently only a single input variable can be optimized via CMA ' + \
            'but got: {}'.format(self.cma_optimizers.keys())

    @torch.no_grad()
    def cma_init(self, var_manager):
        """fresh variables (fresh Adam state) with the grad-free one drawn from the sampler"""
        variables = var_manager.initialize(num_samples=self.num_samples)
        self.sampler.draw(variables, getattr(self, 'shard', None))
        return variables

    @torch.no_grad()
    def cma_update(self, variables, loss=None, inverted_loss=False):
        """tell the sampler `loss`, or (loss=None) the forward-only re-score of `variables`
        -- computed on the un-warped outputs when a transform is being searched"""
        if loss is None:
            loss = self.losses_for_tell(variables) if inverted_loss \
                else np.asarray(self.step(variables, optimize=False)[1])
        self.sampler.report(loss)
        return loss
