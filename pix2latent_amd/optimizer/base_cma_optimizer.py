"""CMA plumbing shared by CMAOptimizer / BasinCMAOptimizer
(reference pix2latent/optimizer/base_cma_optimizer.py:9-215).

Kept semantics: one CMA object per `grad_free` variable (exactly one allowed);
`cma_init` re-initialises ALL variables (fresh Adam state, defaults restored)
and overwrites the grad-free variable with `ask()`; `cma_update` re-scores with
`optimize=False` when no loss is given and tells CMA the ORIGINALLY ASKED
samples with the refined losses (Baldwinian update, :115,140).
With population sharding rank 0 owns the CMA state: its `ask()` is broadcast
and every rank tells the same all-gathered losses, so the replicas stay
identical.
"""
import numpy as np
import torch

from .cma_es import CMAEvolutionStrategy
from ..utils.image import binarize
from ..utils.misc import cprint


class _BaseCMAOptimizer():
    """
    Base template for CMA optimization. Should be used jointly with
    _BaseOptimizer.
    """

    def __init__(self):
        self.num_samples = -1
        self.cma_optimizers = {}
        self._sampled = {}
        self.cma_seed = None
        return

    @torch.no_grad()
    def setup_cma(self, var_manager):
        """ initializes CMA for variables that have the attribute `grad_free` """
        for var_name, var_dict in var_manager.variable_info.items():

            if var_dict['grad_free'] is False:
                continue

            if type(var_dict['grad_free']) == tuple:
                mu, sigma = var_dict['grad_free']
                if mu is None:
                    mu = np.zeros(var_dict['shape'])
                if sigma is None:
                    sigma = 1.
                cma_opt = CMA(mu, sigma=sigma, seed=self.cma_seed)
            else:
                mu = np.zeros(var_dict['shape'])
                cma_opt = CMA(mu, sigma=1.0, seed=self.cma_seed)

            self.cma_optimizers[(var_dict['var_type'], var_name)] = cma_opt
            self.num_samples = max(self.num_samples, cma_opt.batch_size())

        cprint('(cma-es) number of samples: {}'.format(self.num_samples), 'y')

        assert len(self.cma_optimizers.keys()) == 1, \
            'currently only a single input variable can be optimized via CMA ' + \
            'but got: {}'.format(self.cma_optimizers.keys())
        return

    @torch.no_grad()
    def cma_init(self, var_manager):
        """ initializes the provided variable from CMA """
        vars = var_manager.initialize(num_samples=self.num_samples)

        for (var_type, var_name), cma_opt in self.cma_optimizers.items():
            cma_data = cma_opt.ask()
            shard = getattr(self, 'shard', None)
            if shard is not None and shard.enabled:
                cma_data = shard.broadcast_numpy(np.asarray(cma_data), src=0)

            for i, d in enumerate(cma_data):
                leaf = vars[var_type][var_name].data[i]
                leaf.copy_(torch.as_tensor(np.asarray(d), dtype=torch.float32).view_as(leaf))

            self._sampled[(var_type, var_name)] = cma_data

        return vars

    @torch.no_grad()
    def cma_update(self, variables, loss=None, inverted_loss=False):
        """
        Updates the CMA distribution with the provided loss or with a re-score
        of `variables` (forward + loss only).
        """
        for (var_type, var_name), cma_opt in self.cma_optimizers.items():

            cma_data = self._sampled[(var_type, var_name)]

            if loss is None:
                out, loss, _ = self.step(variables, optimize=False)

            if inverted_loss and hasattr(variables, 'transform'):
                target = self.var_manager.variable_info['target']['default']
                weight = self.var_manager.variable_info['weight']['default']

                target = target.unsqueeze(0).type_as(out)
                weight = weight.unsqueeze(0).type_as(out)

                t_fn = self.transform_fns['target']['fn']
                t_param = torch.stack(list(variables.transform.t.data))
                out = t_fn(out, t_param, invert=True)

                loss = self.loss_fn(out, target, binarize(weight))
                loss = loss.cpu().detach().numpy()

            cma_opt.tell(cma_data, np.asarray(loss, dtype=np.float64))
        return loss


class CMA():
    def __init__(self, mu=128 * [0], sigma=1.0, seed=None):
        """
        Wrapper around the CMA-ES strategy with the reference's interface
        (batch_size / ask / tell / mean) and its trick for 1-D problems: the
        variable is duplicated to 2-D with covariance adaptation switched off
        and only the first coordinate is exposed (base_cma_optimizer.py:170-173).
        """
        options = {}
        if seed is not None:
            options['seed'] = seed
        self.is_scalar = False

        mu = np.asarray(mu, dtype=np.float64).reshape(-1)
        if len(mu) == 1:
            mu = np.concatenate([mu, mu])
            options['CMA_on'] = 0
            self.is_scalar = True

        self.cma = CMAEvolutionStrategy(mu, sigma, options)
        return

    def batch_size(self):
        """ Returns the required batch size for CMA """
        return self.cma.sp.popsize

    def ask(self, batch_size=None):
        """ Asks for samples to evaluate. batch_size must be None to train """
        x = np.array(self.cma.ask(batch_size))
        if self.is_scalar:
            self._x = x
            self._x_proxy = x[:, :1]
            return self._x_proxy
        return x

    def tell(self, x, y):
        """ Apply CMA update """
        if self.is_scalar:
            assert x is self._x_proxy or np.array_equal(x, self._x_proxy)
            return self.cma.tell(self._x, y)
        return self.cma.tell(x, y)

    def mean(self):
        """ Returns the mean of the current CMA distribution """
        x = self.cma.mean
        if self.is_scalar:
            return x[:1]
        return x
