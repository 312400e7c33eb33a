"""GradientOptimizer (reference pix2latent/optimizer/gradient_optimizer.py:11-56)."""
import time

from .base_optimizer import _BaseOptimizer
from ..utils.misc import progress_print


class GradientOptimizer(_BaseOptimizer):
    """
    Basic gradient optimizer: `num_samples` candidates, `grad_steps` updates with
    the optimizer defined in the variable manager (Adam by default).
    """

    def __init__(self, *args, **kwargs):
        _BaseOptimizer.__init__(self, *args, **kwargs)
        return

    def optimize(self, num_samples, grad_steps, pbar=None):
        """
        Args
            num_samples (int): number of samples to optimize over
            grad_steps (int): number of gradient descent updates.
            pbar: progress bar such as tqdm or st.progress [Default: None]
        """
        self.losses, self.outs = [], []

        variables = self.var_manager.initialize(num_samples=num_samples)

        t_st = time.time()

        for i in range(grad_steps):
            self.step(variables, optimize=True, transform=(i == 0))

            if pbar is not None:
                pbar.progress(i / grad_steps)

            if self.log:
                if ((i + 1) % self.log_iter == 0) or (i + 1 == grad_steps):
                    self.log_result(variables, i + 1)

            if (i + 1) % self.show_iter == 0:
                t_avg = (time.time() - t_st) / self.show_iter
                progress_print('optimize', i + 1, grad_steps, 'c', t_avg)
                t_st = time.time()

        self.gather_population(variables)

        if self.log:
            return variables, self.outs, self.losses

        return variables, [self._final_grid()], [[grad_steps, {'loss': self.loss}]]
