"""VariableManager: per-sample latent / target bookkeeping.

Host-side mirror of the reference pix2latent/variable_manager.py (register
:83-146, unregister :149-164, edit_variable :167-194, initialize :196-240,
split_vars :16-46, save_variables :49-65) with the same names, arguments,
defaults, assertion messages and returned structure
(`vars.<var_type>.<name>.data[i]`, `.hook_fn`, `vars.opt`, `vars.num_samples`).

MI355X-first differences (data layout, not semantics):
  * all N samples of a variable live in ONE contiguous device buffer
    (`vars.<type>.<name>.buf`, shape [N, *shape]); `data[i]` are views of it.
    The closure feeds contiguous chunk slices to the generator, hooks run as one
    kernel per chunk, and Adam is one fused HIP launch per variable
    (`FusedAdam`, libp2l_hip p2l_adam_step) instead of N*vars tiny param groups;
  * no hard-coded `.cuda()` (reference :217): the device is a constructor
    argument, default = ROCm device if present.
"""
import ctypes as C
import pprint

import numpy as np
import torch
import torch.optim as optim

from . import distribution as dist
from .utils.attrdict import AttrDict as edict


def _default_device():
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


class FusedAdam(object):
    """Adam over the contiguous per-variable buffers; one HIP launch per
    (variable, chunk).  Reproduces torch.optim.Adam (lr per variable,
    betas=(0.9, 0.999), eps=1e-8, per-sample step counters) as instantiated at
    reference variable_manager.py:231-238 and stepped at closure.py:65.

    The per-sample step counters live on the DEVICE (p2l_adam_step_dev): no launch argument
    changes from step to step, which is what lets the whole inner step be captured in a HIP
    graph and replayed (optimizer/base_optimizer.py).  `param_groups` / `state_steps()` keep
    the torch-style view for callers that inspect the optimizer."""

    def __init__(self, entries, betas=(0.9, 0.999), eps=1e-8, recycle=None):
        # entries: list of dict(name, buf, lr, leaves); recycle: a previous FusedAdam whose
        # state tensors are reused (zeroed) when the shapes match - keeps device addresses
        # stable across re-initialisations
        from . import _native as N
        self._N = N
        self._lib = N.lib()
        self.betas, self.eps = betas, eps
        self.entries = {}
        self.param_groups = []
        for e in entries:
            buf = e['buf']
            old = recycle.entries.get(e['name']) if recycle is not None else None
            if old is not None and old['m'].shape == buf.shape and old['m'].device == buf.device:
                m, v, steps = old['m'].zero_(), old['v'].zero_(), old['steps'].zero_()
            else:
                m, v = torch.zeros_like(buf), torch.zeros_like(buf)
                steps = torch.zeros(buf.size(0), dtype=torch.int32, device=buf.device)
            self.entries[e['name']] = dict(buf=buf, lr=e['lr'], m=m, v=v, steps=steps)
            for leaf in e['leaves']:
                self.param_groups.append({'params': [leaf], 'lr': e['lr']})

    def state_key(self):
        """device addresses of everything a captured step writes through this optimizer
        (moments and step counters): part of the HIP-graph key of the inner step.  With
        `reuse_buffers` a re-initialised optimizer keeps them; without, a new optimizer that
        happens to sit on recycled variable addresses is told apart by these."""
        return tuple((n, e['m'].data_ptr(), e['v'].data_ptr(), e['steps'].data_ptr())
                     for n, e in sorted(self.entries.items()))

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g['params']:
                p.grad = None

    def state_steps(self, name):
        """per-sample step counts of variable `name` (host copy)"""
        return self.entries[name]['steps'].cpu().tolist()

    def update(self, name, i0, i1, grad):
        """one Adam step for samples [i0, i1) of variable `name` (they advance in lock-step:
        the step number of sample i0 is used for the whole chunk)."""
        N = self._N
        e = self.entries[name]
        p = e['buf'][i0:i1]
        g = grad.contiguous()
        N.check(self._lib.p2l_adam_step_dev(N.ptr(p), N.ptr(g), N.ptr(e['m'][i0:i1]),
                                            N.ptr(e['v'][i0:i1]), N.i64(p.numel()), N.f32(e['lr']),
                                            N.f32(self.betas[0]), N.f32(self.betas[1]),
                                            N.f32(self.eps),
                                            C.c_void_p(e['steps'][i0:i1].data_ptr()), int(i1 - i0),
                                            N.stream()), 'p2l_adam_step_dev')

    def step(self, closure=None):
        """kept for API compatibility (`vars.opt.step(closure)`); the chunk
        closure applies `update` itself on the fused path."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        return loss


def _slice_var(var_data, lo, hi):
    sub = {'data': var_data.data[lo:hi], 'hook_fn': var_data.hook_fn}
    if 'buf' in var_data and var_data.buf is not None:
        sub['buf'] = var_data.buf[lo:hi]
        sub['offset'] = var_data.get('offset', 0) + lo
    for k in ('grad_free', 'requires_grad', 'name'):
        if k in var_data:
            sub[k] = var_data[k]
    return sub


def slice_vars(vars, lo, hi):
    """view of samples [lo, hi) of a variable dictionary (shares storage)."""
    sub = {}
    for var_type, var_dict in vars.items():
        if var_type in ['opt', 'num_samples']:
            continue
        sub[var_type] = {name: _slice_var(v, lo, hi) for name, v in var_dict.items()}
    sub['opt'] = vars.opt
    sub['num_samples'] = hi - lo
    return edict(sub)


def split_vars(vars, size):
    """ Splits variable dictionary into mini chunks of dictionary
    (contiguous slices of `size` samples, the optimizer object is shared) """
    num_splits = int(np.ceil(vars.num_samples / float(size)))
    return [slice_vars(vars, i * size, min((i + 1) * size, vars.num_samples))
            for i in range(num_splits)]


def save_variables(save_path, variables):
    """np.save of the variable dictionary with tensors moved to CPU
    (reference variable_manager.py:49-65; layout `vars.input.z.data[i]` kept)."""
    out = {}
    for var_type, all_vars in variables.items():
        if var_type == 'opt':
            continue
        if not isinstance(all_vars, dict):
            out[var_type] = all_vars
            continue
        out[var_type] = {}
        for var_name, var_data in all_vars.items():
            d = {k: v for k, v in var_data.items() if k not in ('buf', 'data')}
            d['data'] = [t.detach().cpu().clone() for t in var_data.data]
            out[var_type][var_name] = d
    np.save(save_path, edict(out), allow_pickle=True)
    return


class VariableManager():

    def __init__(self, device=None):
        """ A variable manager that creates variables for optimization """
        self.variable_info = {}
        self.device = torch.device(device) if device is not None else _default_device()
        # reuse_buffers: `initialize()` writes the fresh samples into the SAME device buffers
        # (and Adam state tensors) as the previous call with the same num_samples, instead of
        # allocating new ones.  Device addresses then stay put from one CMA generation to the
        # next, so a captured HIP graph of the inner step stays valid for the whole run.  The
        # optimizers switch it on together with graph execution; note that the `variables` of
        # the previous generation then alias the new ones.
        self.reuse_buffers = False
        self._pool = {}
        return

    def __str__(self):
        fmt = '<Variable Manager>\n{}'
        return fmt.format(pprint.pformat(self.variable_info))

    def register(self,
                 variable_name,
                 shape,
                 var_type,
                 requires_grad=True,
                 default=None,
                 distribution=dist.TruncatedNormalModulo(sigma=1.0, trunc=2.0),
                 optimizer=optim.Adam,
                 learning_rate=0.05,
                 hook_fn=None,
                 grad_free=False,
                 ):
        """
        Registers a variable; the specs are used at `initialize`.  Arguments as
        in the reference (variable_manager.py:83-146).
        """
        if variable_name in self.variable_info:
            print('variable `{}`` already exists.'.format(variable_name))
            return False

        if default is not None:
            msg = 'default and shape must match but got {} vs {}'
            assert tuple(default.size()) == shape, \
                msg.format(list(default.size()), shape)

        self.variable_info[variable_name] = {
            'shape': shape,
            'var_type': var_type,
            'requires_grad': requires_grad,
            'default': default,
            'distribution': distribution,
            'optimizer': optimizer,
            'learning_rate': learning_rate,
            'hook_fn': hook_fn,
            'grad_free': grad_free,
        }
        return True

    def unregister(self, *variable_names):
        for v in variable_names:
            try:
                del self.variable_info[v]
            except KeyError:
                print('no variable named {}'.format(v))
        return

    def edit_variable(self, variable_name, replace_dict):
        if variable_name not in self.variable_info.keys():
            print('variable `{}` does not exist'.format(variable_name))
            return False

        for k, v in replace_dict.items():
            if k not in self.variable_info[variable_name].keys():
                print('variable `{}` has no attribute {}'.format(k, v))
                return False
            self.variable_info[variable_name][k] = v
        return True

    def release_pool(self):
        """forget the pooled device buffers (`reuse_buffers`): the tensors of the last
        `initialize()` then belong to their holder alone, the next call allocates anew"""
        self._pool = {}

    @torch.no_grad()
    def initialize(self, num_samples):
        """
        Materialises `num_samples` samples of every registered variable and a
        fresh optimizer (fresh Adam state on every call, as the reference does
        once per CMA generation).
        """
        vars = {}
        params_to_optimize = []
        fused_entries = []
        spec = None
        all_adam = True

        for v, spec in self.variable_info.items():
            if spec['default'] is not None:
                stacked = spec['default'].detach().unsqueeze(0).expand(
                    num_samples, *spec['default'].shape)
            else:
                stacked = spec['distribution'](num_samples, spec['shape'])
            fresh = stacked.detach()
            if fresh.dtype != torch.float32 and fresh.is_floating_point():
                fresh = fresh.float()
            pooled = self._pool.get((v, num_samples)) if self.reuse_buffers else None
            if pooled is not None and pooled[0].shape == fresh.shape:
                buf, filled_version, filled_from = pooled
                # a constant variable (default-valued, not optimised, not rewritten by a torch
                # op such as apply_transform since it was filled) keeps its contents AND its
                # version: caches keyed on it (the loss' target features) stay valid.
                # Optimised variables are always refilled: Adam updates them through raw
                # pointers, which torch's version counter does not see.
                if spec['requires_grad'] or spec['default'] is None or \
                        filled_from is not spec['default'] or buf._version != filled_version:
                    buf.copy_(fresh)
            else:
                buf = fresh.to(self.device, copy=True).contiguous()
            if self.reuse_buffers:
                self._pool[(v, num_samples)] = (buf, buf._version, spec['default'])
            buf.requires_grad_(False)
            data = [buf[i] for i in range(num_samples)]

            if spec['var_type'] not in vars.keys():
                vars[spec['var_type']] = {}

            vars[spec['var_type']][v] = \
                {'data': data,
                 'buf': buf,
                 'name': v,
                 'hook_fn': spec['hook_fn'],
                 'grad_free': spec['grad_free'],
                 'requires_grad': spec['requires_grad']}

            if not spec['requires_grad']:
                continue

            if spec['optimizer'] is not optim.Adam:
                all_adam = False
            for d in data:
                params_to_optimize.append(
                    {'params': d.requires_grad_(True),
                     'lr': spec['learning_rate']}
                )
            fused_entries.append({'name': v, 'buf': buf, 'lr': spec['learning_rate'],
                                  'leaves': data})

        if all_adam and self.device.type == 'cuda' and len(fused_entries) > 0:
            recycle = self._pool.get(('opt', num_samples)) if self.reuse_buffers else None
            vars['opt'] = FusedAdam(fused_entries, recycle=recycle)
            if self.reuse_buffers:
                self._pool[('opt', num_samples)] = vars['opt']
        else:
            # reference quirk kept: the optimizer CLASS is the one of the LAST
            # registered variable (variable_manager.py:238)
            vars['opt'] = spec['optimizer'](params_to_optimize)
        vars['num_samples'] = num_samples
        return edict(vars)
