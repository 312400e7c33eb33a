"""TransformBasinCMAOptimizer: search over a transformation parameter `t` (CMA-ES, one
generation = one population of candidate transformations) while the latents are refined by
Adam inside each generation and carried from one generation to the next by variable
propagation.

API of reference pix2latent/transform/transform_optimizer.py:20-255 (method and attribute
names, `optimize()` arguments and its three-part result).  The loop is the package's shared
generation driver (optimizer/search_loop.py) with two callbacks; the propagation state is a
small object of its own.

Multi-GPU: the transformation population is sharded like any other.  Propagation needs the
whole population (its running mean moves towards the latent of the BEST candidate, which may
live on another rank), so the propagated rows are all-gathered once per generation, and the
re-sampling noise is drawn on rank 0 and broadcast so that every rank holds the same rows.
"""
import numpy as np
import torch

from ..optimizer.base_optimizer import _BaseOptimizer
from ..optimizer.base_cma_optimizer import _BaseCMAOptimizer
from ..optimizer.search_loop import Generation, StepTicker
from ..utils.image import to_image, to_grid, resize_area


class PropagationState(object):
    """running means of the propagated input variables (`means[name]` -> tensor)"""

    def __init__(self):
        self.names = []
        self.means = {}

    def rows(self, variables, name):
        if name not in variables.input:
            raise RuntimeError(f'variable propagation is set for {name} but '
                               'no such variable was found')
        return variables.input[name]

    def seed_mean(self, name, var):
        # first contact: centre of the current population
        if name not in self.means:
            self.means[name] = torch.stack(list(var.data)).mean(0)

    def pull_towards(self, name, row, beta):
        # beta = 1 forgets the history, beta = 0 never moves
        self.means[name] = (1.0 - beta) * self.means[name] + beta * row

    def resample(self, name, like, spread, renormalize):
        fresh = (self.means[name] + spread * torch.randn_like(like)).data
        if renormalize:
            fresh = (fresh - fresh.mean()) / fresh.std()
        return fresh


class TransformBasinCMAOptimizer(_BaseOptimizer, _BaseCMAOptimizer):

    def __init__(self, *args, **kwargs):
        _BaseOptimizer.__init__(self, *args, **kwargs)
        _BaseCMAOptimizer.__init__(self)
        self._vp = PropagationState()

    # -- propagation bookkeeping (reference names) ---------------------------------------
    @property
    def variables_to_propagate(self):
        return self._vp.names

    @property
    def vp_means(self):
        return self._vp.means

    @vp_means.setter
    def vp_means(self, value):
        self._vp.means = value

    def set_variable_propagation(self, variable_name):
        """ tells optimizer which variable to propagate """
        if variable_name in self._vp.names:
            print(f'variable {variable_name} already exists')
            return
        self._vp.names.append(variable_name)

    def del_variable_propagation(self, variable_name):
        """ stops propagating a variable (the reference's guard is inverted and raises) """
        if variable_name not in self._vp.names:
            print(f'variable {variable_name} is not propagated')
            return
        self._vp.names.remove(variable_name)

    def _collect_rows(self, variables, name):
        """sharded runs: make every rank hold the refined rows of the whole population"""
        if not self.shard.enabled:
            return
        var = variables.input[name]
        n = variables.num_samples
        lo, hi = self.shard.bounds(n)
        local = torch.stack(list(var.data[lo:hi])) if hi > lo else \
            var.data[0].new_zeros((0,) + tuple(var.data[0].shape))
        full = self.shard.all_gather_rows(local, n)
        for i in range(n):
            var.data[i].copy_(full[i])

    @torch.no_grad()
    def update_propagation_variable_statistic(self, variables, ema_beta=0.5):
        """moves each running mean towards the latent of this generation's best candidate
        (ranked by the optimizer's current `loss`)"""
        best = int(np.argmin(np.asarray(self.loss)))
        for name in self._vp.names:
            var = self._vp.rows(variables, name)
            self._collect_rows(variables, name)
            self._vp.seed_mean(name, var)
            self._vp.pull_towards(name, var.data[best], ema_beta)

    @torch.no_grad()
    def propagate_variable(self, variables, curr_iter, total_iter,
                           magnitude=1.0, renormalize=True):
        """re-draws the propagated variables around their running means; the spread decays
        linearly with the progress of the search; `renormalize` standardises each sample
        (keeps the latents from collapsing)"""
        spread = magnitude * (1 - (curr_iter / float(total_iter)))
        for name in self._vp.names:
            var = self._vp.rows(variables, name)
            self._vp.seed_mean(name, var)
            if self.shard.enabled:
                drawn = torch.stack([self._vp.resample(name, row, spread, renormalize)
                                     for row in var.data])
                drawn = self.shard.broadcast_tensor(drawn, src=0)
                for row, new in zip(var.data, drawn):
                    row.copy_(new)
            else:
                for row in var.data:
                    row.copy_(self._vp.resample(name, row, spread, renormalize))

    # -- logging ---------------------------------------------------------------------------
    @torch.no_grad()
    def vis_transform(self, variables):
        """collage of the warped, weighted targets of the current population"""
        warped = torch.stack(list(variables.output.target.data)) * \
            torch.stack(list(variables.output.weight.data))
        picture = to_image(to_grid(warped), cv2_format=False)
        if self.log_resize_factor is not None:
            picture = resize_area(np.array(picture, dtype=np.uint8), self.log_resize_factor)
        self.transform_outs.append(picture)

    def get_candidate(self):
        return self._candidate

    # -- the search ------------------------------------------------------------------------
    def optimize(self, meta_steps, grad_steps, last_grad_steps=None, pbar=None):
        """
        Args
            meta_steps (int): number of CMA generations (the last one is not told)
            grad_steps (int): gradient updates per generation
            last_grad_steps (int): gradient updates of the last generation
                [Default: grad_steps]
            pbar: progress bar such as tqdm or st.progress
        Returns
            variables, ([out grid], [target grid], target of the best candidate), losses
            (with log=True: variables, (outs, transform_outs, target of the best), losses)
        """
        self.setup_cma(self.var_manager)
        self.losses, self.outs, self.transform_outs = [], [], []
        self._best_loss, self._candidate = 999, None
        self._vp.means = {}
        self.transform_tracked = []
        if last_grad_steps is None:
            last_grad_steps = grad_steps
        total_steps = (meta_steps - 1) * grad_steps + last_grad_steps
        scores = {'told': None}

        def after_draw(g, variables):
            if g > 0:
                self.propagate_variable(variables, g, meta_steps)
            self.transform_tracked.append(
                torch.stack(list(variables.transform.t.data)).detach().cpu().clone())

        def after_generation(g, variables, told):
            # the last generation is not told: as in the reference it is ranked by the
            # PREVIOUS generation's un-warped losses (by its own step losses if it is the
            # only generation)
            if told is not None:
                scores['told'] = told
            elif scores['told'] is None:
                scores['told'] = np.asarray(self.loss)
            self.update_propagation_variable_statistic(variables)
            ranked = scores['told']
            if np.min(ranked) < self._best_loss:
                self._best_loss = np.min(ranked)
                self._candidate = \
                    variables.transform.t.data[int(np.argmin(ranked))].detach().cpu()

        ticker = StepTicker(self, total_steps, pbar, mark=grad_steps)
        plan = [Generation(grad_steps, True, True, 0)] * (meta_steps - 1) + \
               [Generation(last_grad_steps, True, False, 0)]
        variables = self.run_generations(plan, self.sampler, ticker, self.num_samples,
                                         after_draw=after_draw,
                                         after_generation=after_generation)

        self.gather_population(variables)
        for name in ('target', 'weight'):
            self._collect_output_rows(variables, name)
        best_target = variables.output.target.data[int(np.argmin(scores['told']))]

        if self.log:
            return variables, (self.outs, self.transform_outs, best_target), self.losses

        target_grid = to_grid(torch.stack(list(variables.output.target.data)).cpu())
        return variables, ([self._final_grid()], [target_grid], best_target), self.loss

    def _collect_output_rows(self, variables, name):
        if not self.shard.enabled or name not in variables.output:
            return
        var = variables.output[name]
        n = variables.num_samples
        lo, hi = self.shard.bounds(n)
        local = torch.stack(list(var.data[lo:hi])) if hi > lo else \
            var.data[0].new_zeros((0,) + tuple(var.data[0].shape))
        full = self.shard.all_gather_rows(local, n)
        for i in range(n):
            var.data[i].copy_(full[i])
