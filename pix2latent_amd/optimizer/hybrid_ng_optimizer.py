"""HybridNevergradOptimizer: ask/tell outer loop with an Adam inner loop (the BasinCMA
scheme with a nevergrad-style sampler and a free population size).

API of reference pix2latent/optimizer/hybrid_ng_optimizer.py:12-81;
budget = meta_steps * grad_steps."""
from .base_optimizer import _BaseOptimizer
from .base_ng_optimizer import _BaseNevergradOptimizer
from .search_loop import Generation, StepTicker


class HybridNevergradOptimizer(_BaseOptimizer, _BaseNevergradOptimizer):

    def __init__(self, method, *args, **kwargs):
        _BaseOptimizer.__init__(self, *args, **kwargs)
        _BaseNevergradOptimizer.__init__(self, method=method)

    def optimize(self, num_samples, meta_steps, grad_steps, last_grad_steps=300, pbar=None):
        """
        Args
            num_samples (int): number of samples to optimize
            meta_steps (int): number of ask/tell updates
            grad_steps (int): gradient updates per ask/tell update
            last_grad_steps (int): gradient updates applied to the last drawn samples
            pbar: progress bar such as tqdm or st.progress
        """
        self.losses, self.outs = [], []
        total_steps = meta_steps * grad_steps + last_grad_steps
        self.setup_ng(self.var_manager, budget=meta_steps * grad_steps)
        ticker = StepTicker(self, total_steps, pbar, mark=grad_steps)
        plan = [Generation(grad_steps, True, True, 1)] * meta_steps + \
               [Generation(last_grad_steps, True, False, 1)]
        variables = self.run_generations(plan, self.sampler, ticker,
                                         self._population(num_samples))
        return self.finish(variables, total_steps)
