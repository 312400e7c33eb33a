"""CMAOptimizer: forward-only CMA-ES search followed by Adam fine-tuning
(reference pix2latent/optimizer/cma_optimizer.py:11-93)."""
import time

from .base_optimizer import _BaseOptimizer
from .base_cma_optimizer import _BaseCMAOptimizer
from ..utils.misc import progress_print


class CMAOptimizer(_BaseOptimizer, _BaseCMAOptimizer):

    def __init__(self, *args, **kwargs):
        _BaseOptimizer.__init__(self, *args, **kwargs)
        _BaseCMAOptimizer.__init__(self)
        return

    def optimize(self, meta_steps, grad_steps=0, pbar=None, num_samples=None):
        """
        Args
            meta_steps (int): number of CMA updates
            grad_steps (int): number of gradient updates to apply after CMA
                optimization. [Default: 0]
            pbar: progress bar such as tqdm or st.progress
            num_samples: must be None
        """
        assert num_samples == None, 'PyCMA optimizer has fixed sample size'

        self.setup_cma(self.var_manager)
        self.losses, self.outs, i = [], [], 0
        total_steps = meta_steps + grad_steps

        t_st = time.time()

        # -- CMA optimization (no gradient descent) -- #
        for _ in range(meta_steps):
            variables = self.cma_init(self.var_manager)

            self.step(variables, optimize=False, transform=False)
            i += 1

            if self.log:
                if (i % self.log_iter == 0) or (i == grad_steps):
                    self.log_result(variables, i)

            self.cma_update(variables, inverted_loss=True)

            if pbar is not None:
                pbar.progress(i / total_steps)
            else:
                if i % self.show_iter == 0:
                    t_avg = (time.time() - t_st) / self.show_iter
                    progress_print('optimize', i, total_steps, 'c', t_avg)
                    t_st = time.time()

        # -- Finetune CMA with ADAM optimization -- #
        variables = self.cma_init(self.var_manager)

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
