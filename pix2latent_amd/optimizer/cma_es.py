"""(mu/mu_w, lambda)-CMA-ES with Hansen's default strategy parameters.

The reference drives the third-party package `cma>=3.0.3` (pycma;
requirements.txt:1, call sites pix2latent/optimizer/base_cma_optimizer.py:2,176,
182,187,202,204,210).  pycma is absent in this environment (and on the GPU
box), so this module provides the same small surface the reference touches --
`CMAEvolutionStrategy(x0, sigma0, opts)`, `.sp.popsize`, `.ask(number)`,
`.tell(X, fitness)`, `.mean`, `.sigma`, option keys 'seed', 'popsize',
'CMA_on' -- implementing the published algorithm (N. Hansen, "The CMA Evolution
Strategy: A Tutorial", 2016).  PARITY UNPINNED against pycma's sampling stream:
pycma is never seeded by the reference either (utils/misc.py:17-18), and tell()
is rank-based, so what matters for parity is the RANKING of the losses fed in.
CPU / numpy only: 128x128 covariance work is microseconds next to a generation
of generator evaluations.
"""
import math
import types

import numpy as np


class CMAEvolutionStrategy(object):
    def __init__(self, x0, sigma0, inopts=None):
        opts = dict(inopts or {})
        self.N = N = len(x0)
        self.mean = np.array(x0, dtype=np.float64).copy()
        self.sigma = float(sigma0)
        self.sigma0 = float(sigma0)
        seed = opts.get('seed', None)
        self.rng = np.random.RandomState(seed)
        lam = int(opts.get('popsize', 4 + int(3 * math.log(N))))
        mu = lam // 2
        w = math.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
        w = w / w.sum()
        mueff = 1.0 / np.sum(w ** 2)
        cma_on = float(opts.get('CMA_on', 1))
        cc = (4 + mueff / N) / (N + 4 + 2 * mueff / N)
        cs = (mueff + 2) / (N + mueff + 5)
        c1 = cma_on * 2 / ((N + 1.3) ** 2 + mueff)
        cmu = cma_on * min(1 - c1, 2 * (mueff - 2 + 1 / mueff) / ((N + 2) ** 2 + mueff))
        damps = 1 + 2 * max(0, math.sqrt((mueff - 1) / (N + 1)) - 1) + cs
        self.sp = types.SimpleNamespace(popsize=lam, mu=mu, weights=w, mueff=mueff, cc=cc,
                                        cs=cs, c1=c1, cmu=cmu, damps=damps)
        self.pc = np.zeros(N)
        self.ps = np.zeros(N)
        self.B = np.eye(N)
        self.D = np.ones(N)
        self.C = np.eye(N)
        self.invsqrtC = np.eye(N)
        self.chiN = math.sqrt(N) * (1 - 1. / (4 * N) + 1. / (21 * N ** 2))
        self.countiter = 0
        self.counteval = 0
        self._eigen_iter = 0
        self.best_x, self.best_f = None, np.inf

    # ------------------------------------------------------------------ ask
    def ask(self, number=None):
        n = self.sp.popsize if number is None else int(number)
        z = self.rng.randn(n, self.N)
        y = (z * self.D) @ self.B.T
        return [self.mean + self.sigma * y[k] for k in range(n)]

    # ----------------------------------------------------------------- tell
    def tell(self, solutions, function_values):
        sp, N = self.sp, self.N
        X = np.asarray([np.asarray(s, dtype=np.float64) for s in solutions])
        f = np.asarray(function_values, dtype=np.float64).reshape(-1)
        if X.shape[0] != f.shape[0]:
            raise ValueError('solutions and function values must have the same length')
        if X.shape[0] < sp.mu:
            raise ValueError('need at least mu=%d solutions' % sp.mu)
        self.countiter += 1
        self.counteval += len(f)
        order = np.argsort(f, kind='stable')
        if f[order[0]] < self.best_f:
            self.best_f, self.best_x = float(f[order[0]]), X[order[0]].copy()
        xold = self.mean
        Xs = X[order[:sp.mu]]
        self.mean = sp.weights @ Xs
        y_w = (self.mean - xold) / self.sigma
        self.ps = (1 - sp.cs) * self.ps + \
            math.sqrt(sp.cs * (2 - sp.cs) * sp.mueff) * (self.invsqrtC @ y_w)
        ps_norm = np.linalg.norm(self.ps)
        hsig = float(ps_norm / math.sqrt(1 - (1 - sp.cs) ** (2 * self.countiter)) / self.chiN
                     < 1.4 + 2. / (N + 1))
        self.pc = (1 - sp.cc) * self.pc + hsig * math.sqrt(sp.cc * (2 - sp.cc) * sp.mueff) * y_w
        if sp.c1 > 0 or sp.cmu > 0:
            Y = (Xs - xold) / self.sigma
            self.C = (1 - sp.c1 - sp.cmu) * self.C + \
                sp.c1 * (np.outer(self.pc, self.pc) + (1 - hsig) * sp.cc * (2 - sp.cc) * self.C) + \
                sp.cmu * (Y.T * sp.weights) @ Y
        self.sigma *= math.exp(min(1.0, (sp.cs / sp.damps) * (ps_norm / self.chiN - 1)))
        # lazy eigendecomposition, O(N^2) amortised
        if (sp.c1 + sp.cmu) > 0 and \
                self.countiter - self._eigen_iter > 1. / (sp.c1 + sp.cmu) / N / 10:
            self._eigen_iter = self.countiter
            self.C = np.triu(self.C) + np.triu(self.C, 1).T
            d2, self.B = np.linalg.eigh(self.C)
            self.D = np.sqrt(np.maximum(d2, 1e-20))
            self.invsqrtC = (self.B / self.D) @ self.B.T

    @property
    def result(self):
        return types.SimpleNamespace(xbest=self.best_x, fbest=self.best_f, xfavorite=self.mean,
                                     evaluations=self.counteval, iterations=self.countiter)

    def stop(self):
        return {}
