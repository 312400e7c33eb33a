"""BasinCMAOptimizer: CMA-ES over the grad-free latent, Adam basin-descent inside.

API of reference pix2latent/optimizer/basincma_optimizer.py:11-83.  Plan: `meta_steps`
generations of `grad_steps` Adam steps each, re-scored and told to the sampler, then one
more draw refined for `last_grad_steps` and not told."""
from .base_optimizer import _BaseOptimizer
from .base_cma_optimizer import _BaseCMAOptimizer
from .search_loop import Generation, StepTicker


class BasinCMAOptimizer(_BaseOptimizer, _BaseCMAOptimizer):

    def __init__(self, *args, **kwargs):
        _BaseOptimizer.__init__(self, *args, **kwargs)
        _BaseCMAOptimizer.__init__(self)

    def optimize(self, meta_steps, grad_steps, last_grad_steps=300, pbar=None,
                 num_samples=None):
        """
        Args
            meta_steps (int): number of CMA updates
            grad_steps (int): number of gradient updates per CMA update.
            last_grad_steps (int): gradient updates applied to the last drawn population.
            pbar: progress bar such as tqdm or st.progress
            num_samples: must be None (the strategy fixes the population size)
        """
        assert num_samples == None, 'PyCMA optimizer has fixed sample size'  # noqa: E711

        self.setup_cma(self.var_manager)
        self.losses, self.outs = [], []
        total_steps = meta_steps * grad_steps + last_grad_steps
        # the reference tests `i + 1` after having incremented i: labels run one ahead
        ticker = StepTicker(self, total_steps, pbar, mark=grad_steps)
        plan = [Generation(grad_steps, True, True, 1)] * meta_steps + \
               [Generation(last_grad_steps, True, False, 1)]
        variables = self.run_generations(plan, self.sampler, ticker, self.num_samples)
        return self.finish(variables, total_steps)
