"""TransformBasinCMAOptimizer: CMA-ES over the transformation parameter with an Adam
inner loop on the latents and variable propagation
(reference pix2latent/transform/transform_optimizer.py:20-255)."""
import time

import numpy as np
import torch

from ..optimizer.base_optimizer import _BaseOptimizer
from ..optimizer.base_cma_optimizer import _BaseCMAOptimizer
from ..utils.misc import progress_print
from ..utils.image import to_image, to_grid, resize_area


class TransformBasinCMAOptimizer(_BaseOptimizer, _BaseCMAOptimizer):
    """
    Transformation search with a BasinCMA-like loop.  Variable propagation
    re-initialises the propagated latent of every new generation around a moving
    average of the best latent found so far, with noise that decays over the
    generations (greatly shortens the inner loops).
    """

    def __init__(self, *args, **kwargs):
        _BaseOptimizer.__init__(self, *args, **kwargs)
        _BaseCMAOptimizer.__init__(self)
        self.variables_to_propagate = []
        return

    @torch.no_grad()
    def vis_transform(self, variables):
        target = torch.stack(list(variables.output.target.data))
        weight = torch.stack(list(variables.output.weight.data))
        transform_im = to_image(to_grid(target * weight), cv2_format=False)
        if self.log_resize_factor is not None:
            transform_im = resize_area(np.array(transform_im, dtype=np.uint8),
                                       self.log_resize_factor)
        self.transform_outs.append(transform_im)
        return

    def set_variable_propagation(self, variable_name):
        """ tells optimizer which variable to propagate """
        if variable_name in self.variables_to_propagate:
            print(f'variable {variable_name} already exists')
            return
        self.variables_to_propagate.append(variable_name)
        return

    def del_variable_propagation(self, variable_name):
        """ stops propagating a variable """
        if variable_name not in self.variables_to_propagate:
            print(f'variable {variable_name} is not propagated')
            return
        self.variables_to_propagate.remove(variable_name)
        return

    def _check_propagated(self, variables, var_name):
        if var_name not in variables.input:
            raise RuntimeError(f'variable propagation is set for {var_name} but '
                               'no such variable was found')

    @torch.no_grad()
    def update_propagation_variable_statistic(self, variables, ema_beta=0.5):
        """ moving average of the propagated variables towards the latent of the
        best candidate of this generation (ema_beta: 1 forgets everything) """
        for var_name in self.variables_to_propagate:
            self._check_propagated(variables, var_name)
            var_data = variables.input[var_name]

            if var_name not in self.vp_means.keys():
                self.vp_means[var_name] = torch.stack(list(var_data.data)).mean(0)

            current_mean = var_data.data[int(np.argmin(np.asarray(self.loss)))]

            self.vp_means[var_name] = \
                ((1.0 - ema_beta) * self.vp_means[var_name]) + (ema_beta * current_mean)
        return

    @torch.no_grad()
    def propagate_variable(self, variables, curr_iter, total_iter,
                           magnitude=1.0, renormalize=True):
        """ resample the propagated variables around the running mean; the noise
        shrinks linearly with the progress; optionally re-standardise each sample """
        for var_name in self.variables_to_propagate:
            self._check_propagated(variables, var_name)
            var_data = variables.input[var_name]

            if var_name not in self.vp_means.keys():
                self.vp_means[var_name] = torch.stack(list(var_data.data)).mean(0)

            z_sigma = magnitude * (1 - (curr_iter / float(total_iter)))

            for i in range(len(var_data.data)):
                _data = (self.vp_means[var_name] +
                         (z_sigma * torch.randn_like(var_data.data[i]))).data
                if renormalize:
                    _data = (_data - _data.mean()) / _data.std()
                var_data.data[i].copy_(_data)
        return

    def get_candidate(self):
        return self._candidate

    def optimize(self, meta_steps, grad_steps, last_grad_steps=None, pbar=None):
        """
        Args
            meta_steps (int): number of CMA updates
            grad_steps (int): number of gradient updates per CMA update.
            pbar: progress bar such as tqdm or st.progress
        """
        self.setup_cma(self.var_manager)
        self.losses, self.outs, self.transform_outs, i = [], [], [], 0
        self._best_loss, self._candidate = 999, None
        self.vp_means = {}
        self.transform_tracked = []

        if last_grad_steps is None:
            last_grad_steps = grad_steps

        total_steps = (meta_steps - 1) * grad_steps + last_grad_steps

        t_st = time.time()
        loss = None

        for meta_iter in range(meta_steps):
            is_last_iter = (meta_iter + 1 == meta_steps)
            _grad_steps = last_grad_steps if is_last_iter else grad_steps

            variables = self.cma_init(self.var_manager)

            if meta_iter > 0:
                self.propagate_variable(variables, meta_iter, meta_steps)

            self.transform_tracked.append(
                torch.stack(list(variables.transform.t.data)).cpu().detach().clone()
            )

            for j in range(_grad_steps):
                self.step(variables, optimize=True, transform=(j == 0))
                i += 1

                if self.log and (j == 0):
                    self.vis_transform(variables)

                if self.log:
                    if (i % self.log_iter == 0) or (i == grad_steps):
                        self.log_result(variables, i)

                if pbar is not None:
                    pbar.progress(i / total_steps)
                else:
                    if i % self.show_iter == 0:
                        t_avg = (time.time() - t_st) / self.show_iter
                        progress_print('optimize', i, total_steps, 'c', t_avg)
                        t_st = time.time()

            if not is_last_iter:
                loss = self.cma_update(variables, inverted_loss=True)
            elif loss is None:                      # meta_steps == 1
                loss = np.asarray(self.loss)

            self.update_propagation_variable_statistic(variables)

            # as in the reference, the last generation re-uses the previous
            # generation's inverted losses here
            if np.min(loss) < self._best_loss:
                self._candidate = \
                    variables.transform.t.data[int(np.argmin(loss))].cpu().detach()
                self._best_loss = np.min(loss)

        candidate_out = variables.output.target.data[int(np.argmin(loss))]

        if self.log:
            return variables, (self.outs, self.transform_outs, candidate_out), \
                self.losses

        transform_target = \
            to_grid(torch.stack(list(variables.output.target.data)).cpu())

        transform_out = self._final_grid()

        results = ([transform_out], [transform_target], candidate_out)

        return variables, results, self.loss
