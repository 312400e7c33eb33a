"""Population sharding over the GPUs of one node.

The candidates of a CMA / Nevergrad generation are independent until the
rank-based `tell`, so the population dimension is block-partitioned across
ranks (one process per GPU, torch.distributed backend 'nccl' = RCCL over xGMI
on the MI355X node, 'gloo' in the CPU tests).  Generator / VGG weights are
replicated once at start-up.  The only traffic is

  * per generation: rank 0's asked population, broadcast ([pop, N] float64, a few KB);
  * per generation: one all-gather of the per-candidate scalar losses (<= 3 floats per
    rank) before `tell`; `optimizer.loss` is a `ShardedLosses` that gathers on first read.
    Reads happen at points every rank reaches in the same order (re-score, log_result with
    log=True, end of optimize()); user code that looks at `opt.loss` must do so on every
    rank, like any collective;
  * at the end: one all-gather of the final latents / images.

All of it is latency-bound scalars / small vectors: no activation, gradient or weight ever
crosses xGMI.  This
replaces the reference's only multi-GPU mechanism, nn.DataParallel over the
StyleGAN2 wrapper (examples/invert_stylegan2_cars_*.py:51-53), which
re-broadcasts all generator weights on every forward.
"""
import numpy as np
import torch
import torch.distributed as dist


from .utils.lazy_losses import LazyLosses


def partition(n, world):
    """contiguous block split of n candidates: first (n % world) ranks get one
    extra.  pop 18 over 8 ranks -> [3,3,2,2,2,2,2,2]."""
    base, extra = divmod(n, world)
    sizes = [base + (1 if r < extra else 0) for r in range(world)]
    bounds, lo = [], 0
    for s in sizes:
        bounds.append((lo, lo + s))
        lo += s
    return bounds


class PopulationShard(object):
    """rank-local view of a population of `n` candidates."""

    def __init__(self, group=None):
        self.group = group
        self.enabled = dist.is_available() and dist.is_initialized() and \
            dist.get_world_size(group) > 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.world = dist.get_world_size(group) if self.enabled else 1

    def bounds(self, n):
        return partition(n, self.world)[self.rank]

    def all_gather_losses(self, local, n):
        """local: 1-D float tensor with this rank's losses -> full [n] tensor in
        population order on every rank (one padded all_gather)."""
        if not self.enabled:
            return local
        parts = partition(n, self.world)
        max_local = max(hi - lo for lo, hi in parts)
        pad = torch.zeros(max_local, dtype=torch.float32, device=local.device)
        pad[:local.numel()] = local.float()
        bufs = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(bufs, pad, group=self.group)
        return torch.cat([bufs[r][:hi - lo] for r, (lo, hi) in enumerate(parts)])

    def broadcast_numpy(self, arr, src=0):
        """rank `src`'s float64 array to every rank (CMA ask)."""
        if not self.enabled:
            return arr
        dev = torch.device('cuda', torch.cuda.current_device()) \
            if dist.get_backend(self.group) == 'nccl' else torch.device('cpu')   # gloo: host buffer
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(dev)
        dist.broadcast(t, src=src, group=self.group)
        return t.cpu().numpy()

    def broadcast_tensor(self, t, src=0):
        """rank `src`'s tensor to every rank (same shape / dtype everywhere); gloo needs a
        host buffer, RCCL a device one"""
        if not self.enabled:
            return t
        nccl = dist.get_backend(self.group) == 'nccl'
        buf = t.detach().contiguous().clone() if (t.is_cuda == nccl) else \
            t.detach().to('cuda' if nccl else 'cpu').contiguous()
        dist.broadcast(buf, src=src, group=self.group)
        return buf.to(t.device)

    def all_gather_rows(self, local, n):
        """gather per-candidate rows (final latents / images) in population order."""
        if not self.enabled:
            return local
        parts = partition(n, self.world)
        max_local = max(hi - lo for lo, hi in parts)
        pad = torch.zeros((max_local,) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
        pad[:local.size(0)] = local
        bufs = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(bufs, pad.contiguous(), group=self.group)
        return torch.cat([bufs[r][:hi - lo] for r, (lo, hi) in enumerate(parts)], dim=0)


class ShardedLosses(LazyLosses):
    """Per-candidate losses of a sharded step.  Holds the rank-local values; the population
    vector is assembled by ONE padded all-gather the first time it is read (`gather()` is
    idempotent and a collective: every rank must reach it)."""

    def __init__(self, local, shard, n):
        LazyLosses.__init__(self, None)
        self.local, self._shard, self._n = local, shard, n

    def gather(self):
        if self._t is None:
            self._t = self._shard.all_gather_losses(self.local, self._n)
        return self._t

    def tensor(self):
        return self.gather()

    def _get(self):
        self.gather()
        return LazyLosses._get(self)

    def __len__(self):
        return self._n
