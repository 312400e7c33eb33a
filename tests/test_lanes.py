"""Host logic of the execution lanes (pix2latent_amd/lanes.py): which lane a thread is in, when a step may
open a second one.  (The device side -- same bits on two streams -- is tests/test_lanes_gpu.py.)"""
import threading

from pix2latent_amd import lanes


class _Ok(object):
    lanes_ok = True


def test_lane_is_per_thread_and_nests():
    assert lanes.current() == 0
    seen = {}

    def other():
        seen['start'] = lanes.current()
        with lanes.use(1):
            seen['inside'] = lanes.current()
        seen['after'] = lanes.current()

    with lanes.use(1):
        assert lanes.current() == 1
        with lanes.use(0):
            assert lanes.current() == 0
        assert lanes.current() == 1
        t = threading.Thread(target=other)
        t.start()
        t.join()
    assert lanes.current() == 0
    assert seen == {'start': 0, 'inside': 1, 'after': 0}


def test_no_second_lane_without_a_device_or_per_lane_scratch(monkeypatch):
    import torch
    monkeypatch.delenv('P2L_STREAMS', raising=False)
    monkeypatch.delenv('P2L_SUBLANES', raising=False)
    if not torch.cuda.is_available():
        assert lanes.wanted(2, _Ok(), _Ok()) == 1          # (CPU tensors: the reference's own sequence)
        assert not lanes.sub_wanted(9, _Ok(), _Ok())
    assert lanes.wanted(2, _Ok(), object()) == 1
    assert lanes.wanted(1, _Ok(), _Ok()) == 1
    monkeypatch.setenv('P2L_STREAMS', '1')
    assert lanes.wanted(4, _Ok(), _Ok()) == 1
    monkeypatch.setenv('P2L_STREAMS', 'not a number')
    assert lanes.wanted(1, _Ok()) == 1


def test_graph_default_follows_the_lanes(monkeypatch):
    """which steps are replayed as a HIP graph unless told otherwise (base_optimizer._graph_default): <= 6 local
    candidates; two reference chunks or more on lanes; one chunk of 7 ... max_batch_size candidates in two
    sub-lanes; NOT an execution pass above the reference chunk, not when P2L_STREAMS=1 leaves one stream."""
    import torch
    from pix2latent_amd.optimizer.base_optimizer import _BaseOptimizer

    class Shard(object):
        enabled = False

    class Stub(object):
        _graph_default = _BaseOptimizer._graph_default
        max_batch_size, exec_batch_size, shard = 9, None, Shard()
        model, loss_fn = _Ok(), type('L', (), {'_engine': _Ok()})()

    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.delenv('P2L_STREAMS', raising=False)
    monkeypatch.delenv('P2L_SUBLANES', raising=False)
    o = Stub()
    assert o._graph_default(3) and o._graph_default(6)
    assert o._graph_default(18)                       # 9 + 9 on two lanes
    assert o._graph_default(9) and o._graph_default(8) and o._graph_default(7)      # one chunk, two sub-lanes
    o.exec_batch_size = 'all'
    assert not o._graph_default(18)                   # one pass of 18 on one stream: eager, as before
    o.exec_batch_size = 18
    assert not o._graph_default(18)
    o.exec_batch_size = None
    monkeypatch.setenv('P2L_STREAMS', '1')
    assert not o._graph_default(18) and not o._graph_default(9) and o._graph_default(6)
    monkeypatch.delenv('P2L_STREAMS')
    o.model = object()                                # no per-lane scratch: no lanes, no graph above 6
    assert not o._graph_default(18) and not o._graph_default(9)
    o.model = _Ok()
    o.shard = type('S', (), {'enabled': True})()
    assert o._graph_default(9)                        # a rank's block of 9: sub-lanes, one graph
    assert not o._graph_default(16)                   # a rank's block of two chunks: lanes, eager
