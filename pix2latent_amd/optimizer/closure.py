"""The inner-loop body: chunk -> hooks -> forward -> loss -> backward -> Adam.

Mirror of the reference pix2latent/optimizer/closure.py:6-79 (`step`), with the
semantics listed in SURVEY.md §3.2 kept:
  * contiguous chunks of `max_batch_size` samples, one optimizer shared by all;
  * per closure call: targets gathered, grads cleared, hooks mutate the latents
    IN PLACE before the forward -- also when `optimize=False` -- then the model
    is called with keyword arguments named after the input variables and the
    loss with keyword arguments named after the output variables;
  * per-sample loss = loss_fn(...).view(b, -1).mean(1); the back-propagated
    scalar is the MEAN over the chunk (gradient factor 1/b_chunk);
  * returns (out [N,3,H,W] detached, per-sample losses, {}).

Two execution paths:
  * fused (default on the ROCm device, `vars.opt` is a FusedAdam): chunk slices of
    the contiguous variable buffers are fed straight to the generator, hooks are
    one kernel per chunk, the gradient comes back as one [b, dim] tensor and
    Adam is one HIP launch per variable; losses stay on the device until someone
    reads them (no per-chunk host sync, reference closure.py:60);
  * generic (any torch optimizer / CPU tensors): the reference's own sequence
    -- stack the per-sample leaves, `opt.step(closure)` -- used by the golden
    trace tests and for user-supplied optimizers.
"""
import os

import numpy as np
import torch

from .. import lanes
from ..variable_manager import split_vars, slice_vars, FusedAdam
from ..utils.lazy_losses import LazyLosses  # noqa: F401  (re-exported)
from ..utils.function_hooks import HookSpan


def _in_sync(var):
    """True when every per-sample tensor still aliases its slot of the
    contiguous buffer (user code may rebind `.data` like the reference does)."""
    buf = var.get('buf', None)
    if buf is None or len(var.data) != buf.size(0):
        return False
    stride = buf.stride(0) * buf.element_size() if buf.size(0) > 1 else 0
    base = buf.data_ptr()
    for i, t in enumerate(var.data):
        if t.data_ptr() != base + i * stride or t.shape != buf.shape[1:]:
            return False
    return True


def _gather(var):
    """[b, *shape] tensor of a variable chunk: zero-copy when possible."""
    if _in_sync(var):
        return var.buf
    return torch.stack(list(var.data))


def apply_hooks(vars, population=None, chunk=None):
    """Runs the hooks of one step: in-place mutation of the latents before the forward pass,
    also on forward-only passes (reference closure.py:42-44).

    The reference calls every hook once per chunk of `max_batch_size` samples, inside that
    chunk's closure.  Here all those calls are made up-front, reference chunk by reference
    chunk over the WHOLE population and for every hooked variable of a chunk in turn -- the
    same call sequence, hence the same random stream, whatever the execution batch size or
    the number of ranks is.  (Safe to hoist: a chunk's hooks touch only that chunk's rows,
    and so does the Adam update the reference interleaves with them.)

    `vars` holds rows [lo, lo + num_samples) of a population of n; population = (lo, n).
    Rows of a chunk held by another rank are passed as absent (utils/function_hooks.py)."""
    lo, n = population if population is not None else (0, vars.num_samples)
    hi = lo + vars.num_samples
    hooked = [v for v in vars.input.values() if v.hook_fn is not None]
    if not hooked:
        return
    with torch.no_grad():
        for c0 in range(0, n, chunk):
            c1 = min(c0 + chunk, n)
            s, e = max(c0, lo), min(c1, hi)
            if s >= e:
                s = e = c0                                 # nothing of this chunk lives here
            for var in hooked:
                if hasattr(var.hook_fn, 'apply_batched') and _in_sync(var):
                    var.hook_fn.apply_batched(var.buf[s - lo:e - lo] if e > s else var.buf[0:0],
                                              HookSpan(s, e, c0, c1))
                elif e > s:
                    var.hook_fn(var.data[s - lo:e - lo])


_SCALES = {}


def _const_scale(n, device):
    """[n] tensor of 1/n on the device, made once (an H2D copy per step would drain the stream)"""
    key = (n, str(device))
    if key not in _SCALES:
        _SCALES[key] = torch.full((n,), 1.0 / n, dtype=torch.float32, device=device)
    return _SCALES[key]


class _LaneCtx(object):
    """chunk ci of a step on lane ci % n: its stream + its workspaces (lanes.py); nothing for n = 1"""

    def __init__(self, n, ci, streams):
        self.on = n > 1
        if self.on:
            self.ctx = (torch.cuda.stream(streams[ci % n]), lanes.use(ci % n))

    def __enter__(self):
        if self.on:
            for c in self.ctx:
                c.__enter__()
        return self

    def __exit__(self, *a):
        if self.on:
            for c in reversed(self.ctx):
                c.__exit__(*a)


def _step_fused(model, vars, loss_fn, optimize, max_batch_size, grad_scale, population):
    outs, losses = [], []
    chunks = split_vars(vars, size=max_batch_size)
    offsets = [ci * max_batch_size for ci in range(len(chunks))]
    engine = getattr(loss_fn, '_engine', None)
    one_pass = population is not None and vars.num_samples > population[2]      # (exec_batch_size: one stream by request)
    if len(chunks) == 1 and not one_pass and lanes.sub_wanted(vars.num_samples, model, engine):
        # one chunk, two lanes (from lanes.SUB_MIN candidates up): rows [0, h) and [h, n), each with the gradient
        # factor of the whole chunk
        n = vars.num_samples
        h = (n + 1) // 2
        if grad_scale is None:
            grad_scale = _const_scale(n, next(iter(vars.input.values())).data[0].device)
        chunks, offsets = [slice_vars(vars, 0, h), slice_vars(vars, h, n)], [0, h]
    split = int(os.environ.get('P2L_LANE_SPLIT', '1') or 1)
    if split > 1 and len(chunks) >= 2 and not one_pass:
        # (measurement, tools/ab_lanes.sh) every reference chunk in `split` parts on as many lanes as P2L_STREAMS
        # allows, each part with the gradient factor of its chunk
        dev0 = next(iter(vars.input.values())).data[0].device
        if grad_scale is None:
            grad_scale = torch.cat([_const_scale(c.num_samples, dev0) for c in chunks])
        parts, offs = [], []
        for ci, c in enumerate(chunks):
            b = c.num_samples
            cuts = [offsets[ci] + (b * j) // split for j in range(split + 1)]
            for lo_, hi_ in zip(cuts[:-1], cuts[1:]):
                if hi_ > lo_:
                    parts.append(slice_vars(vars, lo_, hi_)); offs.append(lo_)
        chunks, offsets = parts, offs
    # the reference chunks of a step are independent (own rows, own Adam update): with more than one they
    # run on side streams, each lane with its own workspaces in the model and the loss -- same bits, the
    # latency-bound layers of one chunk under the busy ones of the other (lanes.py)
    n_lanes = lanes.wanted(len(chunks), model, engine)
    streams, main = None, None
    if n_lanes > 1:
        dev = next(iter(vars.input.values())).data[0].device
        main = torch.cuda.current_stream(dev)
        streams = lanes.side_streams(dev, n_lanes)
        for s in streams:
            s.wait_stream(main)                         # (hooks, tracking copies, the previous step)
    def run_chunk(ci, _vars, n_lanes):
        with _LaneCtx(n_lanes, ci, streams):
            b_sz = _vars.num_samples
            gs = None if grad_scale is None else grad_scale[offsets[ci]: offsets[ci] + b_sz]
            target_args = {k: _gather(v) for k, v in _vars.output.items()}
            leaves, input_args = {}, {}
            for k, var in _vars.input.items():
                x = _gather(var)
                if optimize and var.get('requires_grad', False):
                    x = x.detach().requires_grad_(True)
                    leaves[k] = (x, var)
                input_args[k] = x
            with torch.set_grad_enabled(bool(optimize)):
                out = model(**input_args)
                loss = loss_fn(out, **target_args).view(b_sz, -1).mean(1)
                if optimize:
                    if gs is None:
                        loss.mean().backward()
                    else:
                        (loss * gs).sum().backward()
            if optimize:
                for k, (x, var) in leaves.items():
                    if x.grad is None:
                        continue
                    if not _in_sync(var):
                        raise RuntimeError('variable `%s` was rebound outside its buffer; '
                                           'fused Adam cannot update it' % k)
                    off = var.get('offset', 0)
                    _vars.opt.update(k, off, off + b_sz, x.grad)
            outs.append(out.detach())
            losses.append(loss.detach())
            if n_lanes > 1:                                # (read on the caller's stream after the join)
                outs[-1].record_stream(main)
                losses[-1].record_stream(main)

    for ci, _vars in enumerate(chunks):
        try:
            run_chunk(ci, _vars, n_lanes)
        except torch.cuda.OutOfMemoryError as e:
            if n_lanes == 1 or ci % n_lanes == 0:
                raise
            # The second lane's set of workspaces (generator arena + image staging + loss arena: 5 GB for
            # BigGAN-256 at 9 candidates, 17 GB per lane for StyleGAN2-1024) did not fit beside the first.
            # Nothing of THIS chunk has been updated yet (its Adam update follows its backward), the chunks
            # before it are complete: give the side lanes' memory back, and run this chunk and every later
            # one on lane 0, behind whatever the side streams still hold.
            lanes.give_up('%s in lane %d of %d' % (str(e).split('.')[0], ci % n_lanes, n_lanes))
            for s_ in streams:
                main.wait_stream(s_)
            torch.cuda.current_stream().synchronize()
            lanes.drop_side_scratch(model, engine)
            torch.cuda.empty_cache()
            n_lanes = 1
            run_chunk(ci, _vars, 1)
    if n_lanes > 1:
        for s_ in streams:
            main.wait_stream(s_)
    return torch.cat(outs), LazyLosses(torch.cat(losses)), {}


def _step_generic(model, vars, loss_fn, optimize, max_batch_size, grad_scale, population):
    outs, indiv_losses = [], []
    for ci, _vars in enumerate(split_vars(vars, size=max_batch_size)):
        box = {}
        gs = None if grad_scale is None else \
            grad_scale[ci * max_batch_size: ci * max_batch_size + _vars.num_samples]

        def closure():
            b_sz = _vars.num_samples
            target_args = {k: torch.stack(list(v.data)) for k, v in _vars.output.items()}
            if optimize:
                _vars.opt.zero_grad()
            # (1) hooks: already applied for the whole step (apply_hooks)
            # (2) forward
            input_args = {k: torch.stack(list(v.data)) for k, v in _vars.input.items()}
            out = model(**input_args)
            # (3) loss
            loss = loss_fn(out, **target_args).view(b_sz, -1).mean(1)
            if optimize:
                if gs is None:
                    loss.mean().backward()
                else:
                    (loss * gs).sum().backward()
            box['out'] = out
            box['loss'] = loss.detach().cpu().numpy()

        # (4) optimize
        if optimize:
            _vars.opt.step(closure)
            _vars.opt.zero_grad()
        else:
            with torch.no_grad():
                closure()
        outs.extend(box['out'].detach())
        indiv_losses.extend(box['loss'])
    return torch.stack(outs), indiv_losses, {}


def step(model, vars, loss_fn, optimize=True, max_batch_size=9, grad_scale=None,
         population=None):
    """
    The step function for model evaluation.

    Args:
        vars: variable object generated from variable_manager.
        loss_fn: loss function; called as loss_fn(out, **{output variables}).
        optimize: if False, does not compute gradients nor update anything
            (hooks still run, as in the reference).
        max_batch_size: chunk size.
        grad_scale: optional per-sample gradient weight replacing the implicit
            1/b_chunk of `loss.mean()`; population sharding passes the
            REFERENCE chunk size here so that trajectories do not depend on the
            number of GPUs.
        population: (first row, population size, reference chunk size) when `vars` is a
            block of a larger population and / or `max_batch_size` is an execution batch
            larger than the reference chunk; tells the hooks where their rows sit.

    Returns:
        outs, indiv_losses (list-like of np.float32), misc_return ({})
    """
    apply_hooks(vars, None if population is None else population[:2],
                max_batch_size if population is None else population[2])
    if isinstance(vars.opt, FusedAdam):
        return _step_fused(model, vars, loss_fn, optimize, max_batch_size, grad_scale,
                           population)
    return _step_generic(model, vars, loss_fn, optimize, max_batch_size, grad_scale, population)
