"""GradientOptimizer: `num_samples` candidates refined by `grad_steps` Adam updates.

API of reference pix2latent/optimizer/gradient_optimizer.py:11-56; the loop itself is the
shared driver (search_loop.py) with a one-generation plan and no sampler."""
from .base_optimizer import _BaseOptimizer
from .search_loop import Generation, StepTicker


class GradientOptimizer(_BaseOptimizer):

    def optimize(self, num_samples, grad_steps, pbar=None):
        """
        Args
            num_samples (int): number of samples to optimize over
            grad_steps (int): number of gradient descent updates.
            pbar: progress bar such as tqdm or st.progress [Default: None]
        Returns
            (variables, [image grid], [[grad_steps, {'loss': per-sample losses}]]), or the
            logged lists when constructed with log=True
        """
        self.losses, self.outs = [], []
        # step labels are 1-based counts; the last step is always logged; the pbar is fed
        # the 0-based index and the console line is printed even next to a pbar
        ticker = StepTicker(self, grad_steps, pbar, mark=grad_steps, pbar_lag=1,
                            always_print=True)
        plan = [Generation(steps=grad_steps, refine=True, report=False, shift=0)]
        variables = self.run_generations(plan, None, ticker, num_samples)
        return self.finish(variables, grad_steps)
