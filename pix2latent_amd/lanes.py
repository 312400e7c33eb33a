"""Execution lanes: the reference chunks of one step on separate HIP streams.

The reference evaluates a population chunk by chunk (`max_batch_size` = 9 of 18; reference
pix2latent/optimizer/closure.py:23-27), each chunk forward -> loss -> backward -> Adam on its own rows:
the chunks are independent.  On an MI355X one chunk of 9 leaves the 4^2 ... 16^2 layers of the generator
and the loss network latency-bound (a few dozen blocks on 256 CUs), and so does one pass of 18; two chunks
in flight on two streams fill each other's gaps: 16.4 -> 15.2 ms per step of 18 (tools/two_stream_probe.py).
Such a step is replayed as ONE HIP graph with two branches by default (optimizer/base_optimizer.py): 2 launches
instead of 520 per step on the host.
Results are the same bits as one stream (a candidate's arithmetic does not depend on what runs beside it).

A lane = a stream + its own workspaces in every object that keeps device scratch (the generator's arena
and image staging, the loss's arena); objects advertise `lanes_ok = True` and look the current lane up
here.  Lane 0 on the caller's stream is the only one unless closure.step opens more.
"""
import os
import threading

import torch

_tls = threading.local()         # the current lane is per host thread: two threads may each drive an optimizer
_side = {}


def current():
    return getattr(_tls, 'lane', 0)


class use(object):
    """`with use(k):` -- workspaces of lane k for this thread (no stream switch: closure._LaneCtx pairs it with
    `torch.cuda.stream`).  autograd runs a backward on its own thread: the Functions remember the forward's
    lane and re-enter it there (model/biggan.py `_BigGANFn.backward`)."""

    def __init__(self, lane):
        self.lane = lane

    def __enter__(self):
        self.prev = current()
        _tls.lane = self.lane
        return self

    def __exit__(self, *a):
        _tls.lane = self.prev


_gave_up = []


def give_up(reason):
    """no more lanes in this process (the second set of workspaces did not fit: closure._step_fused)"""
    if not _gave_up:
        import warnings
        warnings.warn('execution lanes switched off for this process: %s' % reason)
    _gave_up.append(reason)


def drop_side_scratch(*objs):
    """free what lanes > 0 allocated in the objects that keep per-lane scratch"""
    for o in objs:
        held = getattr(o, '_lanes', None)
        if isinstance(held, dict):
            for k in [k for k in held if k != 0]:
                del held[k]


def wanted(n_chunks, *objs):
    """number of lanes for a step of n_chunks reference chunks over objs (model, loss): $P2L_STREAMS
    (default 2; 1 = off) when every object has per-lane workspaces.  Also while a HIP graph is being captured:
    the side streams fork from the capturing stream and join it again, the graph gets two branches and the
    replay runs them side by side (measured: 15.3 ms replayed, 15.5 eager, 19.6 replayed on one stream)"""
    try:
        want = int(os.environ.get('P2L_STREAMS', '2'))
    except ValueError:
        want = 2
    if n_chunks < 2 or want < 2 or _gave_up or not torch.cuda.is_available():
        return 1
    if not all(getattr(o, 'lanes_ok', False) for o in objs):
        return 1
    return min(want, n_chunks)


SUB_MIN = 7


def sub_wanted(n, *objs):
    """a step that is ONE chunk of n candidates -- a rank of a 2-GPU run holds 9, the GradientOptimizer example
    8 -- cut in two for two lanes.  The gradient factor stays the one of the whole chunk (closure._step_fused
    passes it explicitly), so the bits do not change (tests/test_sublanes_gpu.py).  Measured
    (profiles/round5_sublanes.txt): 9 candidates 9.92 -> 9.27 ms eager, 9.07 replayed as one graph; 5: 6.57 ->
    6.41; 3 and 2: 2 % SLOWER (there the step is one chain of tiny dependent launches, and two chains side by
    side are as long as one) -- hence from SUB_MIN candidates up.  $P2L_SUBLANES: 0 = never, 1 = from 2 up."""
    mode = os.environ.get('P2L_SUBLANES', '')
    if mode == '0' or n < 2 or (mode != '1' and n < SUB_MIN):
        return False
    return wanted(2, *objs) == 2


def side_streams(device, n):
    """n streams of `device`, made once per process"""
    key = str(device)
    have = _side.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]
