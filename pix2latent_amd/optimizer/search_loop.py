"""One outer-loop driver for every population optimizer of the package.

The reference spells the same loop five times (optimizer/{gradient,cma,basincma,ng,
hybrid_ng}_optimizer.py and transform/transform_optimizer.py).  Here the loop exists once:

    for generation in plan:                       # a `Generation` record
        variables = sampler.draw(...)             # ask (+ broadcast to the other ranks)
        generation hooks (variable propagation ...)
        for j in range(generation.steps):         # Adam steps, or one forward-only scoring
            owner.step(variables, optimize=generation.refine, transform=...)
            ticker.tick(...)                      # logging / progress / pbar bookkeeping
        if generation.report:
            sampler.report(score_for_tell(...))   # rank-based tell

What differs between the optimizers is DATA: the plan (how many generations, how many
steps each, whether they refine, whether they report), the sampler strategy object
(pycma-like generation sampler, nevergrad-like ask/tell sampler, or none), and how step
numbers are labelled in the log.  The behaviours that the golden traces in tests/golden/
pin (what is asked, what `tell` receives, when variables are re-initialised, the step labels
of `losses`) are documented where they are produced.

Multi-GPU: the sampler lives on every rank but only rank 0's draw counts (it is broadcast);
the losses handed to `report` are all-gathered once per generation, so every replica of the
sampler state stays identical.  Nothing else crosses ranks inside a generation.
"""
import os
import time
from collections import namedtuple

import numpy as np
import torch

from ..utils.image import binarize
from ..utils.misc import progress_print

# steps:  number of inner steps of the generation
# refine: True = Adam steps (step(optimize=True), transform applied on the first one);
#         False = forward-only scoring (step(optimize=False, transform=False))
# report: tell the sampler the (re-scored) losses afterwards
# shift:  step-label offset used by the log / progress conditions of that phase
Generation = namedtuple('Generation', 'steps refine report shift')


class StepTicker(object):
    """Counts inner steps over the whole run and does what the reference does after each of
    them: `log_result` on the log cadence, `pbar.progress`, or the periodic console line.

    The label of a step is `done + shift`.  shift = 1 reproduces the BasinCMA / Hybrid /
    fine-tuning phases of the reference (they test and report `i + 1` AFTER incrementing i,
    basincma_optimizer.py:60-73), shift = 0 the forward-only search phases and the transform
    optimizer (cma_optimizer.py:50-62, transform_optimizer.py:214-223)."""

    def __init__(self, owner, total, pbar=None, mark=None, pbar_lag=0, always_print=False):
        self.owner, self.total, self.pbar = owner, total, pbar
        self.mark = mark                    # extra label that is always logged
        self.pbar_lag = pbar_lag            # GradientOptimizer reports i, not i + 1, to pbar
        self.always_print = always_print    # ... and prints even when a pbar is given
        self.done = 0
        self._t0 = time.time()

    def tick(self, variables, shift):
        o = self.owner
        self.done += 1
        label = self.done + shift
        if o.log and (label % o.log_iter == 0 or label == self.mark):
            o.log_result(variables, label)
        if self.pbar is not None:
            self.pbar.progress((self.done - self.pbar_lag) / self.total)
            if not self.always_print:
                return
        if label % o.show_iter == 0:
            now = time.time()
            progress_print('optimize', label, self.total, 'c', (now - self._t0) / o.show_iter)
            self._t0 = now


class PopulationSampler(object):
    """Strategy object for ONE gradient-free variable: owns ask, the broadcast of the asked
    population and tell.  Subclasses implement `_ask(n)` -> (array [n, *shape], handle) and
    `_tell(handle, losses)`."""

    #: population size fixed by the strategy (pycma) or None (nevergrad: caller chooses)
    population = None

    def __init__(self, var_type, var_name):
        self.var_type, self.var_name = var_type, var_name
        self._handle = None

    #: True on a rank whose strategy object is NOT the authoritative one (an installed pycma /
    #: nevergrad on rank != 0): it is neither asked nor told -- an ask without its tell grows pycma's
    #: archive of sent solutions while mean / sigma / stop() freeze, and nevergrad's sequential or
    #: recast optimizers (Powell, SQPCMA) may block or fall back to random points -- and takes rank
    #: 0's population from the broadcast (its shape is known from the variable)
    _replica = False

    def draw(self, variables, shard=None):
        """ask for `variables.num_samples` candidates and write them into the variable"""
        n = variables.num_samples
        if self._replica:
            leaf0 = variables[self.var_type][self.var_name].data[0]
            values, self._handle = np.zeros((n,) + tuple(leaf0.shape), dtype=np.float64), None
        else:
            values, self._handle = self._ask(n)
        values = np.asarray(values, dtype=np.float64)
        if shard is not None and shard.enabled:
            values = shard.broadcast_numpy(values, src=0)
        dst = variables[self.var_type][self.var_name]
        with torch.no_grad():
            for i in range(n):
                leaf = dst.data[i]
                leaf.copy_(torch.as_tensor(values[i], dtype=torch.float32).view_as(leaf))
        return values

    def report(self, losses):
        """Baldwinian update: the strategy sees the candidates it ASKED for with the losses
        their refined versions reached (reference base_cma_optimizer.py:115,140)."""
        self._tell(self._handle, np.asarray(losses, dtype=np.float64))

    def _ask(self, n):
        raise NotImplementedError

    def _tell(self, handle, losses):
        raise NotImplementedError


def find_grad_free(var_manager):
    """[(var_type, name, mu, sigma)] of the variables registered with `grad_free`
    (True, or a (mu, sigma) tuple whose None entries mean zeros / 1.0;
    reference base_cma_optimizer.py:35-55)."""
    found = []
    for name, spec in var_manager.variable_info.items():
        gf = spec['grad_free']
        if gf is False:
            continue
        mu, sigma = gf if type(gf) == tuple else (None, None)
        if mu is None:
            mu = np.zeros(spec['shape'])
        if sigma is None:
            sigma = 1.
        found.append((spec['var_type'], name, mu, sigma))
    return found


class SearchLoopMixin(object):
    """the driver; mixed into _BaseOptimizer (needs .step, .var_manager, .shard, .loss_fn,
    .transform_fns, .log*, .show_iter, .log_result, .gather_population)."""

    def losses_for_tell(self, variables):
        """What the sampler is told: a forward-only re-score of the (refined) population
        (hooks run, nothing is updated); when a `transform` variable type exists the outputs
        are first warped back with each candidate's own parameter and scored against the
        ORIGINAL target under the binarised weight (reference base_cma_optimizer.py:117-138).
        Sharded runs un-warp and score only the rank-local rows and all-gather the result."""
        out, loss, _ = self.step(variables, optimize=False)
        if not hasattr(variables, 'transform') or 'target' not in self.transform_fns:
            return np.asarray(loss)
        n = variables.num_samples
        lo, hi = self.shard.bounds(n) if self.shard.enabled else (0, n)
        info = self.var_manager.variable_info
        dev = variables.transform.t.data[0].device
        if hi > lo:
            rows = self.out_local if self.shard.enabled else out
            t_rows = torch.stack(list(variables.transform.t.data[lo:hi]))
            target = info['target']['default'].unsqueeze(0).type_as(rows)
            weight = binarize(info['weight']['default'].unsqueeze(0).type_as(rows))
            restored = self.transform_fns['target']['fn'](rows, t_rows, invert=True)
            local = self.loss_fn(restored, target, weight).detach().float().reshape(-1)
        else:
            local = torch.zeros(0, dtype=torch.float32, device=dev)
        if self.shard.enabled:
            local = self.shard.all_gather_losses(local, n)
        return local.cpu().numpy()

    def run_generations(self, plan, sampler, ticker, num_samples,
                        after_draw=None, after_generation=None):
        """runs `plan` (list of Generation); returns the last generation's variables.

        Every generation starts from `var_manager.initialize(num_samples)`: fresh tensors,
        defaults restored, FRESH Adam state (reference base_cma_optimizer.py:79,
        base_ng_optimizer.py:104)."""
        variables = None
        # graph execution of the inner step wants stable device addresses across generations
        pooled_before = self.var_manager.reuse_buffers
        if getattr(self, 'use_graph', False) is not False and torch.cuda.is_available():
            lo, hi = self.shard.bounds(num_samples) if self.shard.enabled else (0, num_samples)
            if self.use_graph or os.environ.get('P2L_GRAPH') == '1' or \
                    (os.environ.get('P2L_GRAPH') != '0' and hi > lo and self._graph_default(hi - lo)):
                self.var_manager.reuse_buffers = True
        try:
            for g_idx, gen in enumerate(plan):
                with torch.no_grad():
                    variables = self.var_manager.initialize(num_samples=num_samples)
                    if sampler is not None:
                        sampler.draw(variables, self.shard)
                if after_draw is not None:
                    after_draw(g_idx, variables)
                for j in range(gen.steps):
                    self.step(variables, optimize=gen.refine, transform=(gen.refine and j == 0))
                    if self.log and j == 0 and hasattr(self, 'vis_transform'):
                        self.vis_transform(variables)
                    ticker.tick(variables, gen.shift)
                told = None
                if gen.report and sampler is not None:
                    with torch.no_grad():
                        told = self.losses_for_tell(variables)
                    sampler.report(told)
                if after_generation is not None:
                    after_generation(g_idx, variables, told)
        finally:
            if self.var_manager.reuse_buffers and not pooled_before:
                # the caller gets the last generation's tensors: they leave the pool (a later
                # initialize() / optimize() on the same VariableManager must not overwrite a
                # result the reference returns as independent tensors), and the captured graphs
                # that point into them are retired
                self.var_manager.reuse_buffers = False
                self.var_manager.release_pool()
                self._graphs = {}
        return variables

    def finish(self, variables, total_steps):
        """the `optimize()` return value shared by the five optimizers:
        (variables, [images], [[step, {'loss': ...}], ...])"""
        self.gather_population(variables)
        if self.log:
            return variables, self.outs, self.losses
        return variables, [self._final_grid()], [[total_steps, {'loss': self.loss}]]
