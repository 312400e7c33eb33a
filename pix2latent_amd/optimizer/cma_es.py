"""(mu/mu_w, lambda)-CMA-ES with the defaults `cma.CMAEvolutionStrategy(x0, sigma0)` runs with.

The reference drives the third-party package `cma>=3.0.3` (pycma; requirements.txt:1, call
sites pix2latent/optimizer/base_cma_optimizer.py:2,176,182,187,202,204,210).  pycma is absent
in this environment (and on the GPU box), so this module provides the small surface the
reference touches -- `CMAEvolutionStrategy(x0, sigma0, opts)`, `.sp.popsize`, `.ask(number)`,
`.tell(X, fitness)`, `.mean`, `.sigma`, `.stop()`, `.result`, option keys 'seed', 'popsize',
'CMA_on', 'CMA_active', 'tolx', 'tolfun', 'tolfunhist', 'maxiter', 'maxfevals' --
implementing the published algorithm (N. Hansen, "The CMA Evolution Strategy: A Tutorial",
arXiv:1604.00772, 2016):

  * population 4 + floor(3 ln N), log-rank recombination weights, cumulative step-size
    adaptation, rank-one + rank-mu covariance update, lazy eigendecomposition;
  * ACTIVE covariance update (negative recombination weights for the worse half of the
    population, tutorial eqs. 49-53 and the scaling of section B.2), which is what pycma 3
    does by default (`CMA_active=True`) and hence what base_cma_optimizer.py:176 gets;
    `{'CMA_active': False}` gives the plain update;
  * pycma's default termination tolerances, reported by `stop()` (the reference never calls
    it: its loops run a fixed number of generations).

PARITY UNPINNED against pycma's sampling stream: pycma is never seeded by the reference
either (utils/misc.py:17-18), and tell() is rank-based, so what matters for parity is the
RANKING of the losses fed in.  CPU / numpy only: 128x128 covariance work is microseconds next
to a generation of generator evaluations.
"""
import math
import types

import numpy as np


def recombination_weights(lam, N, c1_of, cmu_of, active=True):
    """(weights[lam], mu, mueff): positive weights sum to 1; negative ones (active CMA) are
    scaled by min(alpha_mu-, alpha_mueff-, alpha_posdef-) of tutorial eqs. 50-53."""
    raw = math.log((lam + 1) / 2.0) - np.log(np.arange(1, lam + 1))
    mu = lam // 2
    pos, neg = raw[:mu], raw[mu:]
    mueff = pos.sum() ** 2 / (pos ** 2).sum()
    w = np.zeros(lam)
    w[:mu] = pos / pos.sum()
    if active and lam > mu and np.abs(neg).sum() > 0:
        c1, cmu = c1_of(mueff), cmu_of(mueff)
        if cmu > 0:
            mueff_neg = neg.sum() ** 2 / (neg ** 2).sum()
            alpha_mu = 1 + c1 / cmu
            alpha_mueff = 1 + 2 * mueff_neg / (mueff + 2)
            alpha_posdef = (1 - c1 - cmu) / (N * cmu)
            w[mu:] = min(alpha_mu, alpha_mueff, alpha_posdef) * neg / np.abs(neg).sum()
    return w, mu, mueff


class CMAEvolutionStrategy(object):
    def __init__(self, x0, sigma0, inopts=None):
        opts = dict(inopts or {})
        self.opts = opts
        self.N = N = len(x0)
        self.mean = np.array(x0, dtype=np.float64).copy()
        self.sigma = float(sigma0)
        self.sigma0 = float(sigma0)
        self.rng = np.random.RandomState(opts.get('seed', None))
        lam = int(opts.get('popsize', 4 + int(3 * math.log(N))))
        cma_on = float(opts.get('CMA_on', 1))
        active = bool(opts.get('CMA_active', True)) and cma_on > 0

        def c1_of(mueff):
            return cma_on * 2 / ((N + 1.3) ** 2 + mueff)

        def cmu_of(mueff):
            return cma_on * min(1 - c1_of(mueff),
                                2 * (mueff - 2 + 1 / mueff) / ((N + 2) ** 2 + mueff))
        w, mu, mueff = recombination_weights(lam, N, c1_of, cmu_of, active)
        # step-size path constant: pycma's CMAAdaptSigmaCSA uses N + mueff + 3 in the denominator
        # (cma/sigma_adaptation.py, as recalled; the tutorial's (mueff + 2) / (N + mueff + 5) is
        # what purecma uses).  The reference runs pycma, so the pycma value it is.
        cs = (mueff + 2) / (N + mueff + 3)
        self.sp = types.SimpleNamespace(
            popsize=lam, mu=mu, weights=w, mueff=mueff, active=active,
            cc=(4 + mueff / N) / (N + 4 + 2 * mueff / N), cs=cs,
            c1=c1_of(mueff), cmu=cmu_of(mueff),
            damps=1 + 2 * max(0, math.sqrt((mueff - 1) / (N + 1)) - 1) + cs)
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
        self._fit_hist = []                 # best fitness of the last generations
        self._last_fit = None
        # pycma's default termination settings
        self.tol = dict(tolx=float(opts.get('tolx', 1e-11)), tolfun=float(opts.get('tolfun', 1e-11)),
                        tolfunhist=float(opts.get('tolfunhist', 1e-12)),
                        maxiter=opts.get('maxiter', 100 + 150 * (N + 3) ** 2 // lam ** 0.5),
                        maxfevals=opts.get('maxfevals', np.inf))

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
        self._fit_hist = (self._fit_hist + [float(f[order[0]])])[-(10 + int(30 * N / sp.popsize)):]
        self._last_fit = f[order]

        xold = self.mean
        Y = (X[order] - xold) / self.sigma                      # ranked steps, best first
        n_used = min(len(Y), sp.popsize)
        w = sp.weights[:n_used]
        self.mean = xold + self.sigma * (w[:sp.mu] @ Y[:sp.mu])
        y_w = (self.mean - xold) / self.sigma

        # cumulation: step-size path (in the isotropic frame) and covariance path
        self.ps = (1 - sp.cs) * self.ps + \
            math.sqrt(sp.cs * (2 - sp.cs) * sp.mueff) * (self.invsqrtC @ y_w)
        ps_norm = np.linalg.norm(self.ps)
        hsig = float(ps_norm / math.sqrt(1 - (1 - sp.cs) ** (2 * self.countiter)) / self.chiN
                     < 1.4 + 2. / (N + 1))
        self.pc = (1 - sp.cc) * self.pc + hsig * math.sqrt(sp.cc * (2 - sp.cc) * sp.mueff) * y_w

        if sp.c1 > 0 or sp.cmu > 0:
            # negative weights act on steps rescaled to the Mahalanobis length sqrt(N), which
            # keeps C positive definite (tutorial eq. 46)
            w_eff = w.copy()
            neg = w < 0
            if neg.any():
                m2 = np.sum((Y[:n_used][neg] @ self.invsqrtC.T) ** 2, axis=1)
                w_eff[neg] = w[neg] * N / np.maximum(m2, 1e-300)
            Yu = Y[:n_used]
            decay = 1 - sp.c1 - sp.cmu * w.sum() + (1 - hsig) * sp.c1 * sp.cc * (2 - sp.cc)
            self.C = decay * self.C + sp.c1 * np.outer(self.pc, self.pc) + \
                sp.cmu * (Yu.T * w_eff) @ Yu
        self.sigma *= math.exp(min(1.0, (sp.cs / sp.damps) * (ps_norm / self.chiN - 1)))

        # lazy eigendecomposition, O(N^2) amortised
        if (sp.c1 + sp.cmu) > 0 and \
                self.countiter - self._eigen_iter > 1. / (sp.c1 + sp.cmu) / N / 10:
            self._eigen_iter = self.countiter
            self.C = np.triu(self.C) + np.triu(self.C, 1).T
            d2, self.B = np.linalg.eigh(self.C)
            self.D = np.sqrt(np.maximum(d2, 1e-20))
            self.invsqrtC = (self.B / self.D) @ self.B.T

    # ------------------------------------------------------------ reporting
    @property
    def result(self):
        return types.SimpleNamespace(xbest=self.best_x, fbest=self.best_f, xfavorite=self.mean,
                                     evaluations=self.counteval, iterations=self.countiter,
                                     stds=self.sigma * np.sqrt(np.diag(self.C)))

    def stop(self):
        """{condition: value} of the termination criteria that are met (pycma's names and
        default tolerances); empty dict = keep going"""
        met = {}
        if self.countiter == 0:
            return met
        t = self.tol
        if self.countiter >= t['maxiter']:
            met['maxiter'] = t['maxiter']
        if self.counteval >= t['maxfevals']:
            met['maxfevals'] = t['maxfevals']
        hist = self._fit_hist
        if len(hist) >= 10 + int(30 * self.N / self.sp.popsize):
            if max(hist) - min(hist) < t['tolfunhist']:
                met['tolfunhist'] = t['tolfunhist']
            if self._last_fit is not None and \
                    max(max(hist), self._last_fit.max()) - min(min(hist), self._last_fit.min()) < t['tolfun']:
                met['tolfun'] = t['tolfun']
        spread = self.sigma * np.sqrt(np.maximum(np.diag(self.C), 0))
        if np.all(spread < t['tolx']) and np.all(self.sigma * np.abs(self.pc) < t['tolx']):
            met['tolx'] = t['tolx']
        if self.D.max() / max(self.D.min(), 1e-300) > 1e7:
            met['conditioncov'] = 1e14
        return met
