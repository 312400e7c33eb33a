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
