"""_BaseOptimizer: shared step / transform / tracking / logging logic.

Mirror of reference pix2latent/optimizer/base_optimizer.py:9-141 (constructor
arguments, register_transform :44-59, apply_transform :62-78, step :81-97,
track :100-107, log_result :123-141) with three MI355X-first changes:
  * the population can be sharded across ranks (parallel.PopulationShard): each
    rank evaluates a contiguous block of candidates, losses are all-gathered;
  * `track()` keeps device clones and only copies to the host when `tracked` is
    read (the reference forces a D2H sync every step, :105-106);
  * `apply_transform` writes into the contiguous variable buffers in place, which
    also invalidates the loss' cached target features.
"""
import os

import numpy as np
import torch

from .. import lanes
from .closure import step, apply_hooks, LazyLosses
from .search_loop import SearchLoopMixin
from ..parallel import PopulationShard, ShardedLosses
from ..utils.image import to_image, to_grid, binarize, resize_area
from ..variable_manager import slice_vars, FusedAdam


class _BaseOptimizer(SearchLoopMixin):
    """ Base template for gradient optimization """

    def __init__(self, model, var_manager, loss_fn, max_batch_size=9,
                 log=False, track_variables=True, exec_batch_size=None, use_graph=None,
                 **kwargs):
        """
        Args
            model (nn.Module): a model to invert
            var_manager (VariableManager): instance of the variable manager
            loss_fn (callable): loss function to compute gradients with
            max_batch_size (int): maximum batch size; larger populations are
                processed in chunks of this size
            exec_batch_size (int): (extension) how many candidates are pushed
                through the device at once.  The reference chunks by
                `max_batch_size` only because of GPU memory; with 288 GB of HBM the
                whole population fits, so the chunk size that DEFINES the
                semantics (gradient factor 1/b_chunk, reference closure.py:58) is
                kept while the execution batch may be larger.  None = the
                reference behaviour (execute chunk by chunk); 'all' = the whole
                (rank-local) population in one pass.
        """
        self.max_batch_size = max_batch_size
        if exec_batch_size is None:
            # unmodified example scripts can opt in from the environment
            env = os.environ.get('P2L_EXEC_BATCH', '').strip().lower()
            if env:
                exec_batch_size = 'all' if env == 'all' else int(env)
        self.exec_batch_size = exec_batch_size
        # (extension) HIP-graph execution of the inner step: after one eager step the whole
        # device side of a step - hooks, generator forward, loss, backward, Adam: 300-470
        # launches - is captured once and REPLAYED as one graph launch.  Pays off where the
        # population shards (2-3 candidates per GPU: 8.3 -> 6.0 ms per step measured, the step
        # is launch-bound there); at 18 candidates the GPU is busy anyway (+1.4 %).
        # None = $P2L_GRAPH ('1' on, '0' off, default: on for <= 6 local candidates).
        self.use_graph = use_graph
        self._graphs = {}
        self.model = model.eval() if hasattr(model, 'eval') else model
        self.var_manager = var_manager
        self.loss_fn = loss_fn
        self.transform_fns = {}

        self.log = log
        self.log_iter = 5
        self.show_iter = 50
        self.log_resize_factor = None
        self.track_variables = track_variables
        self._tracked = {}
        self._track_ring, self._track_stream = {}, None
        self.shard = PopulationShard()
        return

    # -- tracking ------------------------------------------------------------
    # The reference copies every input variable to the host after every step
    # (base_optimizer.py:100-107: a synchronising .cpu() per variable and step).  Same result --
    # `tracked` is a list of CPU tensors [N, *shape] per variable -- without the sync and without
    # holding the history in HBM (round 2 kept a device clone per step: 247 MB per step for the
    # noise maps of StyleGAN2-1024 at population 22): a step's values are snapshot into a small
    # device ring (device-to-device, on the compute stream) and leave for the host on a side
    # stream while the next steps run; a ring slot is waited for only when the copy that used it
    # `TRACK_RING` steps ago has not finished.  HBM cost: TRACK_RING x one step of the variable.
    TRACK_RING = 2
    TRACK_PIN_BYTES = 32 << 20      # larger steps go to pageable host memory (as the reference's do)
    TRACK_PIN_TOTAL = 256 << 20     # ... and so does the history once this much of it is pinned
    # A device-to-host copy into PAGEABLE memory blocks the host until it has happened -- i.e. until the step
    # it snapshots has finished: with a 34 MB noise variable (StyleGAN2-1024, 3 candidates) the GPU then idled
    # 0.6 - 1.2 ms per step behind a host that could not queue the next one (bench C5: 24.7 ms per step where
    # the kernels take 23.5).  History that is not pinned therefore travels through a small ring of pinned
    # staging buffers: the copy off the device is asynchronous, and the staging buffer is copied into the
    # pageable history tensor by the host later -- when its event has fired, at the latest when the ring comes
    # round or `tracked` is read.
    TRACK_STAGE = 2

    @property
    def tracked(self):
        """{variable name: [per-step CPU tensors [N,*shape]]} like the reference"""
        if self._track_stream is not None:
            self._track_stream.synchronize()
        self._track_drain(wait=True)
        return {k: list(v) for k, v in self._tracked.items()}

    def _track_drain(self, wait=False, stage=None):
        """staged copies whose device-to-host leg is done (all of them if `wait`; the one using `stage` in any
        case) -> their pageable history tensors"""
        pending = getattr(self, '_track_pending', None)
        if not pending:
            return
        keep = []
        for ev, st, host in pending:
            if wait or st is stage or ev.query():
                ev.synchronize()
                host.copy_(st)
            else:
                keep.append((ev, st, host))
        self._track_pending = keep

    def _to_host(self, name, src):
        if not src.is_cuda:
            return src.clone()
        if self._track_stream is None:
            self._track_stream = torch.cuda.Stream(device=src.device)
        ring = self._track_ring.setdefault(name, {'slots': [], 'i': 0})
        main = torch.cuda.current_stream(src.device)
        k = ring['i'] % self.TRACK_RING
        ring['i'] += 1
        if k < len(ring['slots']) and ring['slots'][k]['dev'].shape == src.shape:
            slot = ring['slots'][k]
            main.wait_event(slot['done'])           # its previous copy to the host has left
        else:
            slot = {'dev': torch.empty_like(src), 'done': torch.cuda.Event()}
            if k < len(ring['slots']):
                # the replaced snapshot may still be on its way to the host: its block must not
                # return to the allocator of the compute stream before that copy has read it
                ring['slots'][k]['dev'].record_stream(self._track_stream)
                main.wait_event(ring['slots'][k]['done'])
                ring['slots'][k] = slot
            else:
                ring['slots'].append(slot)
        slot['dev'].copy_(src)                      # snapshot: later steps overwrite `src`
        ready = torch.cuda.Event()
        ready.record(main)
        nbytes = src.numel() * src.element_size()
        pinned = getattr(self, '_track_pinned', 0)
        pin = nbytes <= self.TRACK_PIN_BYTES and pinned + nbytes <= self.TRACK_PIN_TOTAL
        if pin:
            self._track_pinned = pinned + nbytes
        host = torch.empty(src.shape, dtype=src.dtype, pin_memory=pin)
        dst = host
        if not pin:
            stages = ring.setdefault('stage', [])
            j = ring.setdefault('stage_i', 0) % self.TRACK_STAGE
            ring['stage_i'] = j + 1
            if j >= len(stages) or stages[j].shape != src.shape:
                st = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
                if j < len(stages):
                    self._track_drain(stage=stages[j])
                    stages[j] = st
                else:
                    stages.append(st)
            self._track_drain(stage=stages[j])          # (its copy of TRACK_STAGE steps ago, and whatever else is done)
            dst = stages[j]
        with torch.cuda.stream(self._track_stream):
            self._track_stream.wait_event(ready)
            dst.copy_(slot['dev'], non_blocking=True)
            slot['done'].record(self._track_stream)
            if dst is not host:
                ev = torch.cuda.Event()
                ev.record(self._track_stream)
                if not hasattr(self, '_track_pending'):
                    self._track_pending = []
                self._track_pending.append((ev, dst, host))
        return host

    def track(self, variables):
        for v_name, v_data in variables.input.items():
            buf = v_data.get('buf', None) if hasattr(v_data, 'get') else None
            src = buf if buf is not None else torch.stack(list(v_data.data))
            self._tracked.setdefault(v_name, []).append(self._to_host(v_name, src.detach()))
        return

    def register_benchmark(self, benchmark):
        self.bm = benchmark
        return

    # -- transforms ----------------------------------------------------------
    def register_transform(self, transform_fn, tranform_var_name, target_var_name):
        """
        Applies transformation function using the transform_var on the target
        variables before optimizing.
        """
        self.transform_fns[target_var_name] = {
            'fn': transform_fn,
            'transform_param': tranform_var_name,
            'target_var': target_var_name
        }
        return

    @torch.no_grad()
    def apply_transform(self, variables, transform_dict):
        t_fn = transform_dict['fn']
        src_name = transform_dict['transform_param']
        dst_name = transform_dict['target_var']

        src_type = self.var_manager.variable_info[src_name]['var_type']
        dst_type = self.var_manager.variable_info[dst_name]['var_type']

        # a rank only ever looks at its own block of candidates: warp just those rows
        n = variables.num_samples
        lo, hi = self.shard.bounds(n) if self.shard.enabled else (0, n)
        if hi == lo:
            return
        src_data = torch.stack(list(variables[src_type][src_name].data[lo:hi]))
        dst_data = torch.stack(list(variables[dst_type][dst_name].data[lo:hi]))

        new_dst_data = t_fn(dst_data, src_data)
        for i in range(len(new_dst_data)):
            variables[dst_type][dst_name].data[lo + i].copy_(new_dst_data[i])
        return

    # -- the step --------------------------------------------------------------
    def _grad_scale(self, n, lo, hi, device):
        """1 / (size of the REFERENCE chunk each candidate would sit in).  Cached: building
        it with torch.tensor(list, device=...) is a pageable H2D copy that synchronises the
        stream, i.e. one full GPU drain per step (cost 2.5 ms of idle GPU per 26 ms step)."""
        key = (n, lo, hi, str(device), self.max_batch_size)
        cache = self.__dict__.setdefault('_gs_cache', {})
        if key not in cache:
            mbs = self.max_batch_size
            sizes = [min(mbs, n - (i // mbs) * mbs) for i in range(lo, hi)]
            cache[key] = torch.tensor([1.0 / s for s in sizes], dtype=torch.float32, device=device)
        return cache[key]

    def step(self, variables, optimize=True, transform=False):
        """one pass over the population: (transform on request) -> track -> closure.step on
        this rank's block of candidates, executed `exec_batch_size` at a time"""
        if len(self.transform_fns) > 0 and transform:
            for _, transform_dict in self.transform_fns.items():
                self.apply_transform(variables, transform_dict)

        if self.track_variables:
            self.track(variables)

        n = variables.num_samples
        mbs = self.max_batch_size
        ebs = n if self.exec_batch_size == 'all' else self.exec_batch_size
        sharded = self.shard.enabled
        lo, hi = self.shard.bounds(n) if sharded else (0, n)
        first = next(iter(variables.input.values())).data[0]

        graphed = self._graphed_step(variables, optimize, transform, lo, hi)
        if graphed is not None:
            return graphed

        if not sharded and (ebs is None or ebs <= mbs or n <= mbs):
            # the reference's own execution: chunk by chunk, gradient factor 1/b_chunk implied
            self.out, self.loss, self.other = step(self.model, variables, loss_fn=self.loss_fn,
                                                   optimize=optimize, max_batch_size=mbs)
            return self.out, self.loss, self.other

        if hi == lo:                      # a rank without candidates (population < world)
            self._idle_hooks(variables, n)
            out, self.other = None, {}
            loss_t = torch.zeros(0, dtype=torch.float32, device=first.device)
        else:
            block = slice_vars(variables, lo, hi) if sharded else variables
            out, loss, self.other = step(
                self.model, block, loss_fn=self.loss_fn, optimize=optimize,
                max_batch_size=mbs if ebs is None else max(mbs, ebs),
                grad_scale=self._grad_scale(n, lo, hi, first.device), population=(lo, n, mbs))
            if not sharded:
                self.out, self.loss = out, loss
                return self.out, self.loss, self.other
            loss_t = loss.tensor() if isinstance(loss, LazyLosses) else \
                torch.tensor(np.asarray(loss), dtype=torch.float32, device=first.device)

        self.out_local = self.out = out
        # ONE all-gather per generation, not per step: `ShardedLosses` gathers
        # when it is read -- re-score before tell, log_result, end of optimize() -- points
        # every rank reaches in the same order.  Re-scores and log=True read it right away.
        self.loss = ShardedLosses(loss_t, self.shard, n)
        if self.log or not optimize:
            self.loss.gather()
        return self.out, self.loss, self.other

    # -- HIP-graph execution ---------------------------------------------------------------
    def _graph_wanted(self, variables, local_n):
        if self.use_graph is None:
            env = os.environ.get('P2L_GRAPH', '').strip()
            if env in ('0', '1'):
                self.use_graph = env == '1'
        if self.use_graph is False or not isinstance(variables.opt, FusedAdam):
            return False
        if not torch.cuda.is_available() or self.log:
            return False
        for var in variables.input.values():
            # a hook is replayed verbatim: its strength must not be a host number that changes
            if var.hook_fn is not None and not getattr(var.hook_fn, 'graph_safe', False):
                return False
        return True if self.use_graph else self._graph_default(local_n)

    def _graph_default(self, local_n):
        """graph replay of the inner step unless told otherwise: where the step is launch-bound (<= 6
        local candidates) and where it runs as lanes on several streams (two reference chunks or more:
        520 launches per step of 18 on the host otherwise, and one graph with two branches replays
        faster than the eager streams)"""
        if local_n <= 6:
            return True
        ebs = self.exec_batch_size
        chunked = ebs is None or (ebs != 'all' and ebs <= self.max_batch_size)
        n_chunks = -(-local_n // self.max_batch_size) if chunked else 1
        objs = (self.model, getattr(self.loss_fn, '_engine', None))
        if n_chunks >= 2:
            return (not self.shard.enabled) and lanes.wanted(n_chunks, *objs) > 1
        # one chunk: two lanes inside it from lanes.SUB_MIN candidates up, unless it is an execution pass
        # larger than the reference chunk (exec_batch_size: one stream by request)
        return local_n <= self.max_batch_size and lanes.sub_wanted(local_n, *objs)

    def _graphed_step(self, variables, optimize, transform, lo, hi):
        """optimize steps without a transform, on variables whose device buffers were seen
        before: replay the captured graph (capturing it on the second sighting).  Returns
        None when the step has to run eagerly."""
        if not optimize or transform or hi == lo or not self._graph_wanted(variables, hi - lo):
            return None
        # identity of everything the captured launches point at: the variable buffers, and for
        # the output variables (targets, weights) also their version - the loss keeps target
        # features cached per version, a transform that rewrites the targets must re-capture
        # ... and WHICH optimizer / workspaces they belong to: the caching allocator hands the
        # addresses of freed buffers out again (vm.initialize() in a loop gives new variables
        # and a new Adam state at the old addresses), and a model or loss that re-allocates its
        # workspace for a larger batch leaves the old pointers baked into the captured launches
        key = (variables.num_samples, lo, hi, self.max_batch_size, self.exec_batch_size,
               variables.opt.state_key(), getattr(self.model, 'ws_generation', 0),
               getattr(getattr(self.loss_fn, '_engine', None), 'generation', 0)) + tuple(
            (name, v.buf.data_ptr()) for name, v in sorted(variables.input.items())
            if v.get('buf', None) is not None) + tuple(
            (name, v.buf.data_ptr(), v.buf._version) for name, v in sorted(variables.output.items())
            if v.get('buf', None) is not None)
        entry = self._graphs.get(key)
        if entry is None:                 # first sighting: run eagerly (this is the warm-up)
            self._graphs = {key: 'warm'}  # (a new set of buffers retires the old graphs)
            return None
        if entry == 'warm':
            try:
                graph = torch.cuda.CUDAGraph()
                saved = (self.use_graph, self.track_variables)
                self.use_graph, self.track_variables = False, False   # (tracking stays eager)
                try:
                    with torch.cuda.graph(graph):
                        out, loss, other = self.step(variables, optimize=True, transform=False)
                finally:
                    self.use_graph, self.track_variables = saved
            except Exception as e:       # capture is an optimisation: fall back, loudly, once
                import warnings
                warnings.warn('HIP-graph capture of the inner step failed (%s: %s); running '
                              'eagerly' % (type(e).__name__, e))
                self.use_graph = False
                self._graphs = {}
                return None
            # the entry PINS what the graph points at (variables -> buffers, Adam moments and step
            # counters): none of those addresses can be recycled while the graph can be replayed
            entry = self._graphs[key] = (graph, out, loss, other, variables, variables.opt)
        graph, out, loss, other = entry[:4]
        graph.replay()
        # The captured tensors are static buffers every replay refills.  What the step hands out are COPIES
        # (one 14 MB device copy per step of 18 x 256^2, ~5 us; 18 floats of losses): the reference returns
        # independent tensors per step (closure.py:68-79), and a caller -- or a LazyLosses read after a later
        # step -- that keeps `opt.out` / `opt.loss` of step i must not see step i+1 in them.
        self.out = self.out_local = out.clone()
        if isinstance(loss, ShardedLosses):
            self.loss = ShardedLosses(loss.local.clone(), self.shard, variables.num_samples)
        else:
            self.loss = LazyLosses(loss.tensor().clone()) if isinstance(loss, LazyLosses) else loss
        self.other = other
        return self.out, self.loss, self.other

    def _idle_hooks(self, variables, n):
        """keep an idle rank's random stream and hook clocks in step with the others"""
        apply_hooks(slice_vars(variables, 0, 0), (0, n), self.max_batch_size)

    def gather_population(self, variables):
        """after sharded optimisation: make every rank's `variables` and
        `self.out` hold the whole population again (final return value)."""
        if not self.shard.enabled:
            return
        n = variables.num_samples
        lo, hi = self.shard.bounds(n)
        if isinstance(self.loss, ShardedLosses):
            self.loss.gather()
        with torch.no_grad():
            for _, v in variables.input.items():
                full = self.shard.all_gather_rows(torch.stack(list(v.data[lo:hi])) if hi > lo
                                                  else v.data[0].new_zeros((0,) + tuple(v.data[0].shape)), n)
                for i in range(n):
                    v.data[i].copy_(full[i])
            if self.out_local is not None:
                proto = self.out_local
            else:
                proto = None
            shape = self.shard_out_shape(proto, variables)
            local = proto if proto is not None else torch.zeros((0,) + shape, device=v.data[0].device)
            self.out = self.shard.all_gather_rows(local, n)

    def shard_out_shape(self, proto, variables=None):
        """shape of one output image on a rank that owns no candidate (population < world
        size): the generator renders at the target's size, whatever the model"""
        if proto is not None:
            self._out_shape = tuple(proto.shape[1:])
        elif not hasattr(self, '_out_shape') and variables is not None and 'target' in variables.output:
            self._out_shape = tuple(variables.output.target.data[0].shape)
        return getattr(self, '_out_shape', (3, 256, 256))

    def optimize(self):
        raise NotImplementedError

    def benchmark(self, variables, out):
        """quality metrics through a registered benchmark object (the reference
        version, base_optimizer.py:114-120, raises NameError)."""
        t_var = variables.get('transform', {}).get('t', None) if hasattr(variables, 'get') else None
        if t_var is not None and 'target' in self.transform_fns:
            out = self.transform_fns['target']['fn'](out, torch.stack(list(t_var.data)), invert=True)
        target = variables.output.target.data[0].unsqueeze(0)
        weight = binarize(variables.output.weight.data[0].clone()).unsqueeze(0)
        return self.bm.evaluate(out, target, weight)

    def log_result(self, variables, step_iter):
        if hasattr(self, 'bm'):
            res = self.benchmark(variables, self.out)
        else:
            res = {'loss': np.array(self.loss)}
        self.losses.append([step_iter, res])

        collage = to_image(to_grid(self.out.cpu()), cv2_format=False)

        if self.log_resize_factor is not None:
            collage = resize_area(np.array(collage, dtype=np.uint8), self.log_resize_factor)

        self.outs.append(collage)
        return

    def _final_grid(self):
        return to_grid(torch.stack(list(self.out.cpu().detach())))
