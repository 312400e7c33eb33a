"""CMAOptimizer: forward-only CMA-ES search, then Adam fine-tuning of a last draw.

API of reference pix2latent/optimizer/cma_optimizer.py:11-93.  Plan: `meta_steps`
generations of ONE forward-only scoring step (no transform, nothing updated) told to the
sampler, then one more draw refined for `grad_steps`."""
from .base_optimizer import _BaseOptimizer
from .base_cma_optimizer import _BaseCMAOptimizer
from .search_loop import Generation, StepTicker


class CMAOptimizer(_BaseOptimizer, _BaseCMAOptimizer):

    def __init__(self, *args, **kwargs):
        _BaseOptimizer.__init__(self, *args, **kwargs)
        _BaseCMAOptimizer.__init__(self)

    def optimize(self, meta_steps, grad_steps=0, pbar=None, num_samples=None):
        """
        Args
            meta_steps (int): number of CMA updates
            grad_steps (int): gradient updates applied after the search [Default: 0]
            pbar: progress bar such as tqdm or st.progress
            num_samples: must be None (the strategy fixes the population size)
        """
        assert num_samples == None, 'PyCMA optimizer has fixed sample size'  # noqa: E711

        self.setup_cma(self.var_manager)
        self.losses, self.outs = [], []
        total_steps = meta_steps + grad_steps
        ticker = StepTicker(self, total_steps, pbar, mark=grad_steps)
        plan = [Generation(1, False, True, 0)] * meta_steps + \
               [Generation(grad_steps, True, False, 1)]
        variables = self.run_generations(plan, self.sampler, ticker, self.num_samples)
        return self.finish(variables, total_steps)
