"""Execution lanes (pix2latent_amd/lanes.py): the reference chunks of a step -- independent by construction,
/root/reference/pix2latent/optimizer/closure.py:23-79 runs them one after the other -- on two HIP streams,
each lane with its own generator arena, image staging and loss arena.  The bits are those of one stream
(tests/test_shard_bits_gpu.py compares every step of the bench problem); here: the lanes exist, are
distinct, are not opened for one chunk, under P2L_STREAMS=1, or for objects without per-lane scratch."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _problem(dev, n):
    import warnings
    warnings.simplefilter('ignore')
    from pix2latent_amd import VariableManager, distribution
    from pix2latent_amd.utils import synthetic as S, function_hooks as hook
    from pix2latent_amd.model.biggan import BigGAN
    from pix2latent_amd.optimizer import GradientOptimizer
    import pix2latent_amd.loss_functions as LF
    torch.manual_seed(3)
    model = BigGAN(weights=S.biggan_weights(0), device=dev)
    loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=S.lpips_vgg_weights(1), device=dev)
    vm = VariableManager(device=dev)
    vm.register('z', (128,), 'input', distribution=distribution.TruncatedNormalModulo(), learning_rate=0.05,
                hook_fn=hook.Clamp(2.0))
    vm.register('c', (128,), 'input', default=0.05 * torch.randn(128), learning_rate=0.01)
    vm.register('target', (3, 256, 256), 'output', requires_grad=False, default=S.synthetic_target(256, 1))
    vm.register('weight', (3, 256, 256), 'output', requires_grad=False, default=S.synthetic_weight_mask(256))
    opt = GradientOptimizer(model, vm, loss_fn, max_batch_size=3)
    return opt, vm.initialize(num_samples=n), model, loss_fn


def _run(dev, n, steps=2):
    opt, variables, model, loss_fn = _problem(dev, n)
    losses = []
    for i in range(steps):
        opt.step(variables, optimize=True, transform=(i == 0))
        losses.append(torch.as_tensor([float(x) for x in opt.loss]))
    opt.step(variables, optimize=False)
    losses.append(torch.as_tensor([float(x) for x in opt.loss]))
    torch.cuda.synchronize()
    return model, loss_fn._engine, torch.stack(losses), variables.input.z.buf.clone()


def test_two_chunks_two_lanes_same_bits(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    monkeypatch.setenv('P2L_STREAMS', '2')
    model, eng, l2, z2 = _run(dev, 7)                     # chunks 3, 3, 1 on lanes 0, 1, 0
    assert sorted(model._lanes) == [0, 1] and sorted(eng._lanes) == [0, 1]
    a, b = model._lanes[0], model._lanes[1]
    assert a.ws.data_ptr() != b.ws.data_ptr() and a.img16.data_ptr() != b.img16.data_ptr()
    assert eng._lanes[0].ws.data_ptr() != eng._lanes[1].ws.data_ptr()
    monkeypatch.setenv('P2L_STREAMS', '1')
    model1, eng1, l1, z1 = _run(dev, 7)
    assert sorted(model1._lanes) == [0] and sorted(eng1._lanes) == [0]
    assert torch.equal(l1, l2), 'losses on two streams differ from one stream'
    assert torch.equal(z1, z2), 'latents after two Adam steps differ'


def test_chunks_cut_into_parts_on_more_lanes_same_bits(monkeypatch):
    """P2L_LANE_SPLIT (the measurement hook of profiles/round6_lane_split.txt): every reference chunk in parts, each
    with the gradient factor of its chunk, on up to P2L_STREAMS lanes -- the bits of one stream"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    monkeypatch.setenv('P2L_STREAMS', '1')
    monkeypatch.delenv('P2L_LANE_SPLIT', raising=False)
    _, _, l1, z1 = _run(dev, 7)
    monkeypatch.setenv('P2L_STREAMS', '3')
    monkeypatch.setenv('P2L_LANE_SPLIT', '2')
    model, eng, l3, z3 = _run(dev, 7)                     # chunks 3, 3, 1 -> parts 1, 2, 1, 2, 1 on three lanes
    assert sorted(model._lanes) == [0, 1, 2] and sorted(eng._lanes) == [0, 1, 2]
    assert torch.equal(l1, l3), 'losses of the parts differ from the chunks on one stream'
    assert torch.equal(z1, z3), 'latents after two Adam steps differ'


def test_one_chunk_opens_no_lane(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    monkeypatch.setenv('P2L_STREAMS', '2')
    model, eng, _, _ = _run(torch.device('cuda:0'), 3, steps=1)
    assert sorted(model._lanes) == [0] and sorted(eng._lanes) == [0]


def test_wanted_needs_every_object():
    from pix2latent_amd import lanes

    class Plain(object):
        pass

    class Ok(object):
        lanes_ok = True
    if torch.cuda.is_available():
        assert lanes.wanted(2, Ok(), Ok()) == 2
        assert lanes.wanted(3, Ok(), Ok()) == 2
    assert lanes.wanted(2, Ok(), Plain()) == 1
    assert lanes.wanted(1, Ok(), Ok()) == 1
    assert lanes.wanted(2, Ok(), None) == 1


def test_second_lane_out_of_memory_falls_back_to_one_lane(monkeypatch):
    """the two-lane default doubles the generator and loss arenas; when the SECOND set does not
    fit (StyleGAN2-1024 at 9 candidates: 17 GB per lane) the step must not die.  The first allocation of
    lane 1's generator arena is made to fail: the chunk is re-run on lane 0, the side scratch is freed,
    lanes stay off for the process, and the bits are those of one stream."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from pix2latent_amd import lanes
    from pix2latent_amd.model import biggan as BG
    dev = torch.device('cuda:0')
    monkeypatch.setenv('P2L_STREAMS', '1')
    _, _, l1, z1 = _run(dev, 7)
    monkeypatch.setenv('P2L_STREAMS', '2')
    monkeypatch.setattr(lanes, '_gave_up', [])
    real_ws, fired = BG.BigGAN._workspace, []

    def failing_workspace(self, B):
        if lanes.current() == 1 and not fired:
            fired.append(1)
            raise torch.cuda.OutOfMemoryError('HIP out of memory (injected). Tried to allocate 3.4 GiB')
        return real_ws(self, B)
    monkeypatch.setattr(BG.BigGAN, '_workspace', failing_workspace)
    model, eng, l2, z2 = _run(dev, 7)
    assert fired and lanes._gave_up and 'out of memory' in lanes._gave_up[0]
    assert lanes.wanted(2, model, eng) == 1
    assert sorted(model._lanes) == [0] and sorted(eng._lanes) == [0]
    assert torch.equal(l1, l2) and torch.equal(z1, z2)


def test_replayed_step_hands_out_independent_tensors(monkeypatch):
    """a HIP-graph replay refills static buffers; what `opt.out` / `opt.loss` hold after a step are
    copies, so a caller that keeps step i's results does not find step i+1 in them (the reference returns
    fresh tensors per step, closure.py:68-79)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    monkeypatch.setenv('P2L_STREAMS', '2')
    opt, variables, model, loss_fn = _problem(dev, 6)
    opt.use_graph = True
    kept = []
    for i in range(5):
        opt.step(variables, optimize=True, transform=(i == 0))
        kept.append((opt.out, opt.loss, opt.out.clone(), torch.as_tensor([float(x) for x in opt.loss])))
    torch.cuda.synchronize()
    assert any(isinstance(v, tuple) for v in opt._graphs.values()), 'the step was not replayed from a graph'
    for out, loss, out_then, loss_then in kept:
        assert torch.equal(out, out_then)
        assert torch.equal(torch.as_tensor([float(x) for x in loss]), loss_then)
    assert not torch.equal(kept[-1][0], kept[-2][0])
