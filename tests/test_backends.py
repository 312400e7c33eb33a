"""optimizer/backends.py: the installed `cma` / `nevergrad` packages are preferred (they are what
the reference instantiates: base_cma_optimizer.py:2,176, base_ng_optimizer.py:1,81-83), the
in-tree restatements are the fallback.  Both branches, with recording fakes standing in for the
packages (neither is installed in the build environment)."""
import importlib
import os
import sys
import types

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _toy import FakeCMAES, FakeNGOpt, fake_nevergrad  # noqa: E402


@pytest.fixture
def fresh(monkeypatch):
    """reload the modules that resolve a backend at import, and put them back afterwards"""
    names = ['pix2latent_amd.optimizer.backends', 'pix2latent_amd.optimizer.base_cma_optimizer',
             'pix2latent_amd.optimizer.base_ng_optimizer']

    def reload_all():
        return [importlib.reload(importlib.import_module(n)) for n in names]
    yield reload_all
    monkeypatch.undo()
    for k in ('cma', 'nevergrad'):
        sys.modules.pop(k, None) if isinstance(sys.modules.get(k), types.ModuleType) and \
            getattr(sys.modules.get(k), '_p2l_fake', False) else None
    reload_all()


def _fake_module(name, **attrs):
    m = types.ModuleType(name)
    m._p2l_fake = True
    m.__version__ = '0.0-fake'
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def test_fallback_when_packages_are_absent(fresh, monkeypatch):
    for k in ('cma', 'nevergrad'):
        monkeypatch.setitem(sys.modules, k, None)          # import raises ImportError
    B, C, G = fresh()
    from pix2latent_amd.optimizer import cma_es, ng_compat
    assert C.CMAEvolutionStrategy is cma_es.CMAEvolutionStrategy and 'in-tree' in C.CMA_BACKEND
    assert G.ng is ng_compat and not G.NG_EXTERNAL
    es = C.CMA(mu=np.zeros(6), sigma=0.5, seed=3)
    x = es.ask()
    assert x.shape == (es.batch_size(), 6)
    es.tell(x, np.arange(len(x), dtype=np.float64))


def test_installed_packages_are_preferred(fresh, monkeypatch):
    ng = fake_nevergrad()
    monkeypatch.setitem(sys.modules, 'cma', _fake_module('cma', CMAEvolutionStrategy=FakeCMAES))
    monkeypatch.setitem(sys.modules, 'nevergrad', _fake_module('nevergrad', optimizers=ng.optimizers, p=ng.p))
    monkeypatch.delenv('P2L_SAMPLERS', raising=False)     # (tests/conftest.py pins the in-tree samplers)
    B, C, G = fresh()
    assert C.CMA_EXTERNAL
    assert C.CMAEvolutionStrategy is FakeCMAES and C.CMA_BACKEND.startswith('pycma')
    assert G.NG_EXTERNAL and G.NG_BACKEND.startswith('nevergrad')
    # the facade drives the package exactly as the reference does: x0, sigma0, options
    FakeCMAES.log = []
    es = C.CMA(mu=np.arange(4.0), sigma=0.7)
    asked = es.ask()
    es.tell(asked, np.ones(len(asked)))
    assert FakeCMAES.log and np.array_equal(FakeCMAES.log[-1][0], asked)
    # nevergrad proper takes no seed argument: the random state hangs off the parametrisation
    FakeNGOpt.log, FakeNGOpt.instances = [], []
    s = G.AskTellSampler('input', 'z', 'CMA', np.zeros(5), budget=12, seed=None)
    assert FakeNGOpt.instances[-1].budget == 12 and s.opt is FakeNGOpt.instances[-1]


def test_environment_forces_the_in_tree_samplers(fresh, monkeypatch):
    monkeypatch.setitem(sys.modules, 'cma', _fake_module('cma', CMAEvolutionStrategy=FakeCMAES))
    monkeypatch.setenv('P2L_SAMPLERS', 'intree')
    B, C, G = fresh()
    from pix2latent_amd.optimizer import cma_es
    assert C.CMAEvolutionStrategy is cma_es.CMAEvolutionStrategy


def test_replicas_of_an_external_backend_are_not_told(fresh, monkeypatch):
    """sharded run with an INSTALLED pycma: rank 0's strategy is the only one that is read (its ask
    is broadcast), a replica would tell solutions pycma never sent from there -- it is not told"""
    monkeypatch.setitem(sys.modules, 'cma', _fake_module('cma', CMAEvolutionStrategy=FakeCMAES))
    monkeypatch.delenv('P2L_SAMPLERS', raising=False)
    B, C, G = fresh()

    class Shard(object):
        enabled = True
        def __init__(self, rank): self.rank = rank
        def broadcast_numpy(self, a, src=0): return a

    import torch

    for rank, told in ((0, True), (1, False)):
        FakeCMAES.log = []
        s = C.PycmaSampler('input', 'z', np.zeros(4), 1.0)
        n = s.population
        leaves = [torch.zeros(4) for _ in range(n)]
        class V(dict):
            num_samples = n
        v = V(input={'z': types.SimpleNamespace(data=leaves)})
        asks = []
        real_ask = s.es.ask
        s.es.ask = lambda *a, **k: (asks.append(1), real_ask(*a, **k))[1]
        for _ in range(3):                       # three generations
            s.draw(v, Shard(rank))
            s.report(np.arange(n, dtype=np.float64))
        assert bool(FakeCMAES.log) == told
        # a replica is not ASKED either (an ask without its tell grows pycma's archive of
        # sent solutions and leaves a sequential nevergrad optimizer waiting)
        assert len(asks) == (3 if told else 0)


def test_replicas_of_an_external_nevergrad_are_neither_asked_nor_told(fresh, monkeypatch):
    """the same rule for the ask/tell sampler, with a backend that RAISES on an ask that follows an
    un-told ask (what nevergrad's sequential optimizers effectively do)"""
    class Strict(FakeNGOpt):
        pending = 0
        def ask(self):
            assert type(self).pending < self.num_workers_allowed, 'ask without the matching tell'
            type(self).pending += 1
            return FakeNGOpt.ask(self)
        def tell(self, cand, value):
            type(self).pending -= 1
            return FakeNGOpt.tell(self, cand, value)
        num_workers_allowed = 4
    mod = _fake_module('nevergrad', optimizers=types.SimpleNamespace(registry={'CMA': Strict}),
                       p=fake_nevergrad().p)
    monkeypatch.setitem(sys.modules, 'nevergrad', mod)
    monkeypatch.delenv('P2L_SAMPLERS', raising=False)
    B, C, G = fresh()
    import torch

    class Shard(object):
        enabled = True
        def __init__(self, rank): self.rank = rank
        def broadcast_numpy(self, a, src=0): return a

    for rank in (0, 1):
        Strict.pending = 0
        FakeNGOpt.log = []
        s = G.AskTellSampler('input', 'z', 'CMA', np.zeros(5), budget=40, seed=None)
        leaves = [torch.zeros(5) for _ in range(4)]
        class V(dict):
            num_samples = 4
        v = V(input={'z': types.SimpleNamespace(data=leaves)})
        for _ in range(3):
            s.draw(v, Shard(rank))
            s.report(np.arange(4, dtype=np.float64))
        assert Strict.pending == 0
        kinds = [e[0] for e in FakeNGOpt.log]
        assert kinds.count('tell') == (12 if rank == 0 else 0) and kinds.count('ask') == kinds.count('tell')
