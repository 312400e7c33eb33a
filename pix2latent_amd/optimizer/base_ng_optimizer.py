"""nevergrad-style ask/tell sampler for NevergradOptimizer / HybridNevergradOptimizer.

API kept from reference pix2latent/optimizer/base_ng_optimizer.py (:27-171): constructor
argument `method` with the validity assertion, `setup_ng(var_manager, budget)`,
`ng_init(var_manager, num_samples)`, `ng_update(variables, loss, inverted_loss)`, attribute
`ng_optimizers`.  The third-party `nevergrad` package is replaced by optimizer/ng_compat.py
(PARITY UNPINNED); the module-level name `ng` is what the golden-trace tests patch.

Unlike pycma the number of candidates per round is the caller's choice: `n` single asks
that may straddle the strategy's own generations, and one tell per candidate.
"""
import numpy as np
import torch

from . import backends
from .search_loop import PopulationSampler, find_grad_free
from ..utils.misc import cprint

# the installed nevergrad when it imports (what the reference calls, base_ng_optimizer.py:1,81-83),
# optimizer/ng_compat.py otherwise
ng, NG_BACKEND, NG_EXTERNAL = backends.nevergrad()

#: methods that evaluate one candidate at a time (not an exhaustive list, as in the reference)
SEQUENTIAL_METHODS = ('SQPCMA', 'chainCMAPowell', 'Powell')


class AskTellSampler(PopulationSampler):

    def __init__(self, var_type, var_name, method, mu, budget, seed=None):
        PopulationSampler.__init__(self, var_type, var_name)
        # the reference leaves the mutation sigma at its default of 1 (its set_mutation call
        # is commented out, base_ng_optimizer.py:82)
        self.opt = backends.make_ng_optimizer(ng, NG_EXTERNAL, method, mu, budget, seed=seed)

    def _ask(self, n):
        candidates = [self.opt.ask() for _ in range(n)]
        # every `.args` is the 1-tuple (array,): concatenation stacks them to [n, *shape]
        return np.concatenate([c.args for c in candidates]), candidates

    def draw(self, variables, shard=None):
        # (as PycmaSampler: with an INSTALLED nevergrad only rank 0's optimizer is authoritative --
        #  an unseeded replica asks different candidates than the ones rank 0 broadcasts)
        self._replica = bool(NG_EXTERNAL and shard is not None and shard.enabled and shard.rank != 0)
        return PopulationSampler.draw(self, variables, shard)

    def _tell(self, candidates, losses):
        if getattr(self, '_replica', False):
            return
        for cand, value in zip(candidates, losses):
            self.opt.tell(cand, float(value))


class _BaseNevergradOptimizer(object):
    """mixin used together with _BaseOptimizer"""

    def __init__(self, method):
        self.method = method
        self.valid_methods = list(ng.optimizers.registry.keys())
        self.sequential_methods = list(SEQUENTIAL_METHODS)
        self.is_sequential = method in SEQUENTIAL_METHODS
        if self.is_sequential:
            cprint('{} is a sequential method. batch size is set to 1'.format(method), 'y')
        assert method in self.valid_methods, f'unknown nevergrad method: {self.method}'
        self.ng_optimizers = {}       # {(var_type, name): AskTellSampler}
        self.ng_seed = None

    @property
    def sampler(self):
        return next(iter(self.ng_optimizers.values()))

    def _population(self, num_samples):
        return 1 if self.is_sequential else num_samples

    @torch.no_grad()
    def setup_ng(self, var_manager, budget):
        """one ask/tell optimizer per `grad_free` variable; exactly one is supported"""
        for var_type, name, mu, _sigma in find_grad_free(var_manager):
            self.ng_optimizers[(var_type, name)] = \
                AskTellSampler(var_type, name, self.method, mu, budget, seed=self.ng_seed)

        assert len(self.ng_optimizers.keys()) == 1, \
            'currently only a single input variable can be optimized via ' + \
            'Nevergrad but got: {}'.format(self.ng_optimizers.keys())

    @torch.no_grad()
    def ng_init(self, var_manager, num_samples):
        variables = var_manager.initialize(num_samples=self._population(num_samples))
        self.sampler.draw(variables, getattr(self, 'shard', None))
        return variables

    @torch.no_grad()
    def ng_update(self, variables, loss=None, inverted_loss=False):
        if loss is None:
            loss = self.losses_for_tell(variables) if inverted_loss \
                else np.asarray(self.step(variables, optimize=False)[1])
        self.sampler.report(loss)
