"""NevergradOptimizer: forward-only ask/tell search + Adam fine-tuning
(reference pix2latent/optimizer/ng_optimizer.py:12-91)."""
import time

from .base_optimizer import _BaseOptimizer
from .base_ng_optimizer import _BaseNevergradOptimizer
from ..utils.misc import progress_print


class NevergradOptimizer(_BaseOptimizer, _BaseNevergradOptimizer):

    def __init__(self, method, *args, **kwargs):
        _BaseOptimizer.__init__(self, *args, **kwargs)
        _BaseNevergradOptimizer.__init__(self, method=method)
        return

    def optimize(self, num_samples, meta_steps, grad_steps=0, pbar=None):
        """
        Args
            num_samples (int): number of samples per ask/tell round
            meta_steps (int): number of ask/tell updates
            grad_steps (int): gradient updates applied after the search
        """
        self.losses, self.outs, i = [], [], 0
        total_steps = meta_steps + grad_steps
        self.setup_ng(self.var_manager, budget=meta_steps)

        t_st = time.time()

        for _ in range(meta_steps):
            variables = self.ng_init(self.var_manager, num_samples)
            self.step(variables, optimize=False, transform=False)
            i += 1

            if self.log:
                if (i % self.log_iter == 0) or (i == grad_steps):
                    self.log_result(variables, i)

            self.ng_update(variables, inverted_loss=True)

            if pbar is not None:
                pbar.progress(i / total_steps)
            else:
                if i % self.show_iter == 0:
                    t_avg = (time.time() - t_st) / self.show_iter
                    progress_print('optimize', i, total_steps, 'c', t_avg)
                    t_st = time.time()

        variables = self.ng_init(self.var_manager, num_samples)

        for j in range(grad_steps):
            self.step(variables, optimize=True, transform=(j == 0))
            i += 1

            if self.log:
                if ((i + 1) % self.log_iter == 0) or (i + 1 == grad_steps):
                    self.log_result(variables, i + 1)

            if pbar is not None:
                pbar.progress(i / total_steps)
            else:
                if (i + 1) % self.show_iter == 0:
                    t_avg = (time.time() - t_st) / self.show_iter
                    progress_print('optimize', i + 1, total_steps, 'c', t_avg)
                    t_st = time.time()

        self.gather_population(variables)

        if self.log:
            return variables, self.outs, self.losses

        return variables, [self._final_grid()], [[total_steps, {'loss': self.loss}]]
