"""Which evolution-strategy packages drive the gradient-free outer loops.

The reference instantiates `cma.CMAEvolutionStrategy` (pix2latent/optimizer/
base_cma_optimizer.py:2,176) and `nevergrad.optimizers.registry[method]`
(base_ng_optimizer.py:1,81-83).  A drop-in must give a user of the reference the SAME sampler
numerics when those packages are installed, so they are preferred whenever they import; the
in-tree restatements (cma_es.py: CMA-ES with pycma-3's defaults and active covariance update;
ng_compat.py: the ask / tell facade) are the fallback for environments without them -- this
build environment and the GPU box, where neither is installed (PARITY UNPINNED for the
fallback).  `P2L_SAMPLERS=intree` forces the fallback (read here, once).
"""
import importlib
import os


def _want_external():
    return os.environ.get('P2L_SAMPLERS', '').strip().lower() not in ('intree', 'in-tree', 'builtin')


def cma_strategy():
    """-> (CMAEvolutionStrategy class, description)"""
    if _want_external():
        try:
            cma = importlib.import_module('cma')
            return cma.CMAEvolutionStrategy, 'pycma %s' % getattr(cma, '__version__', '?')
        except ImportError:
            pass
    from .cma_es import CMAEvolutionStrategy
    return CMAEvolutionStrategy, 'in-tree CMA-ES (optimizer/cma_es.py)'


def nevergrad():
    """-> (module with `.p.Array` and `.optimizers.registry`, description, is_external)"""
    if _want_external():
        try:
            ng = importlib.import_module('nevergrad')
            return ng, 'nevergrad %s' % getattr(ng, '__version__', '?'), True
        except ImportError:
            pass
    from . import ng_compat
    return ng_compat, 'in-tree ask/tell facade (optimizer/ng_compat.py)', False


def make_ng_optimizer(ng, is_external, method, mu, budget, seed=None):
    """the reference's construction (base_ng_optimizer.py:81-83): an Array parametrisation
    around `mu`, mutation sigma left at its default.  nevergrad proper takes no seed argument:
    its random state hangs off the parametrisation."""
    factory = ng.optimizers.registry[method]
    param = ng.p.Array(init=mu)
    if not is_external:
        return factory(parametrization=param, budget=budget, seed=seed)
    opt = factory(parametrization=param, budget=budget)
    if seed is not None:
        opt.parametrization.random_state.seed(seed)
    return opt
