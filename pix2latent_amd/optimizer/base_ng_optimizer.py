"""Nevergrad-style ask/tell plumbing shared by NevergradOptimizer and
HybridNevergradOptimizer (reference pix2latent/optimizer/base_ng_optimizer.py:
setup_ng :52-91, ng_init :94-118, ng_update :121-171).  The third-party
`nevergrad` package is replaced by optimizer/ng_compat.py (parity unpinned)."""
import numpy as np
import torch

from . import ng_compat as ng
from ..utils.image import binarize
from ..utils.misc import cprint


class _BaseNevergradOptimizer():
    """
    Base template for ask/tell black-box optimisation; used jointly with
    _BaseOptimizer.  `num_samples` is free (asks may straddle generations).
    """

    def __init__(self, method):
        self.method = method
        self.valid_methods = [x[0] for x in ng.optimizers.registry.items()]

        # this is not an exhaustive list
        self.sequential_methods = ['SQPCMA', 'chainCMAPowell', 'Powell']
        self.is_sequential = self.method in self.sequential_methods

        if self.is_sequential:
            seq_msg = '{} is a sequential method. batch size is set to 1'
            cprint(seq_msg.format(self.method), 'y')

        assert self.method in self.valid_methods, \
            f'unknown nevergrad method: {self.method}'

        self.ng_optimizers = {}
        self._sampled = {}
        self.ng_seed = None
        return

    @torch.no_grad()
    def setup_ng(self, var_manager, budget):
        """ one ask/tell optimizer per `grad_free` variable (exactly one allowed) """
        for var_name, var_dict in var_manager.variable_info.items():

            if var_dict['grad_free'] is False:
                continue

            if type(var_dict['grad_free']) == tuple:
                mu, sigma = var_dict['grad_free']
                if mu is None:
                    mu = np.zeros(var_dict['shape'])
                if sigma is None:
                    sigma = 1.
            else:
                mu = np.zeros(var_dict['shape'])
                sigma = 1.0

            opt_fn = ng.optimizers.registry[self.method]
            p = ng.p.Array(init=mu)  # the reference leaves the mutation sigma at 1
            ng_opt = opt_fn(parametrization=p, budget=budget, seed=self.ng_seed)

            self.ng_optimizers[(var_dict['var_type'], var_name)] = ng_opt

        assert len(self.ng_optimizers.keys()) == 1, \
            'currently only a single input variable can be optimized via ' + \
            'Nevergrad but got: {}'.format(self.ng_optimizers.keys())
        return

    @torch.no_grad()
    def ng_init(self, var_manager, num_samples):
        if self.is_sequential:
            num_samples = 1
        vars = var_manager.initialize(num_samples=num_samples)

        for (var_type, var_name), ng_opt in self.ng_optimizers.items():
            ng_data = [ng_opt.ask() for _ in range(num_samples)]
            # every args is a 1-tuple (array,): stacks to [num_samples, *shape] (reference :107)
            _ng_data = np.concatenate([x.args for x in ng_data])
            shard = getattr(self, 'shard', None)
            if shard is not None and shard.enabled:
                _ng_data = shard.broadcast_numpy(_ng_data, src=0)

            for i, d in enumerate(_ng_data):
                leaf = vars[var_type][var_name].data[i]
                leaf.copy_(torch.as_tensor(np.asarray(d), dtype=torch.float32).view_as(leaf))

            self._sampled[(var_type, var_name)] = ng_data

        return vars

    @torch.no_grad()
    def ng_update(self, variables, loss=None, inverted_loss=False):
        """ tell every asked candidate its (re-scored) loss """
        for (var_type, var_name), ng_opt in self.ng_optimizers.items():

            ng_data = self._sampled[(var_type, var_name)]

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

            for d, l in zip(ng_data, np.asarray(loss)):
                ng_opt.tell(d, float(l))

        return
