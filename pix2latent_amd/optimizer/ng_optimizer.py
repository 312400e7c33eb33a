"""NevergradOptimizer: forward-only ask/tell search with a free number of candidates per
round, then Adam fine-tuning of a last draw.

API of reference pix2latent/optimizer/ng_optimizer.py:12-91; budget = meta_steps."""
from .base_optimizer import _BaseOptimizer
from .base_ng_optimizer import _BaseNevergradOptimizer
from .search_loop import Generation, StepTicker


class NevergradOptimizer(_BaseOptimizer, _BaseNevergradOptimizer):

    def __init__(self, method, *args, **kwargs):
        _BaseOptimizer.__init__(self, *args, **kwargs)
        _BaseNevergradOptimizer.__init__(self, method=method)

    def optimize(self, num_samples, meta_steps, grad_steps=0, pbar=None):
        """
        Args
            num_samples (int): candidates per ask/tell round
            meta_steps (int): number of ask/tell rounds
            grad_steps (int): gradient updates applied after the search [Default: 0]
            pbar: progress bar such as tqdm or st.progress
        """
        self.losses, self.outs = [], []
        total_steps = meta_steps + grad_steps
        self.setup_ng(self.var_manager, budget=meta_steps)
        ticker = StepTicker(self, total_steps, pbar, mark=grad_steps)
        plan = [Generation(1, False, True, 0)] * meta_steps + \
               [Generation(grad_steps, True, False, 1)]
        variables = self.run_generations(plan, self.sampler, ticker,
                                         self._population(num_samples))
        return self.finish(variables, total_steps)
