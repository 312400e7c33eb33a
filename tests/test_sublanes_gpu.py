"""Sub-lanes (pix2latent_amd/lanes.py sub_wanted): a step that is ONE chunk -- a rank of a 2-GPU run holds 9
candidates (/root/reference/examples/invert_stylegan2_cars_basincma.py:51-53 is the split that is replaced), the
GradientOptimizer example 8 -- cut in two for two streams, gradient factor of the whole chunk kept: same bits.
On by default from 7 candidates up (below, two chains of tiny launches side by side are as long as one);
P2L_SUBLANES=1 forces it from 2 up, =0 switches it off."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize('n,graph,force', [(2, '0', True), (3, '0', True), (3, '1', True), (5, '1', True),
                                           (9, '0', False), (8, '1', False)])
def test_one_chunk_in_two_lanes_same_bits(monkeypatch, n, graph, force):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from tests.test_lanes_gpu import _problem
    dev = torch.device('cuda:0')

    def run(sub):
        if sub and not force:
            monkeypatch.delenv('P2L_SUBLANES', raising=False)    # (the default: on from 7 candidates)
        else:
            monkeypatch.setenv('P2L_SUBLANES', '1' if sub else '0')
        monkeypatch.setenv('P2L_GRAPH', graph)             # ('1': the forked chunk captured as one graph, replayed on step 3)
        opt, variables, model, loss_fn = _problem(dev, n)
        opt.max_batch_size = 9                         # (one chunk)
        losses = []
        for i in range(3):
            opt.step(variables, optimize=True, transform=(i == 0))
            losses.append(torch.as_tensor([float(x) for x in opt.loss]))
        opt.step(variables, optimize=False)
        losses.append(torch.as_tensor([float(x) for x in opt.loss]))
        torch.cuda.synchronize()
        return sorted(model._lanes), torch.stack(losses), variables.input.z.buf.clone(), variables.input.c.buf.clone()
    lanes1, l1, z1, c1 = run(False)
    lanes2, l2, z2, c2 = run(True)
    assert lanes1 == [0] and lanes2 == [0, 1]
    assert torch.equal(l1, l2), 'losses of the two-lane chunk differ'
    assert torch.equal(z1, z2) and torch.equal(c1, c2), 'latents after three Adam steps differ'


def test_small_chunks_stay_on_one_stream_by_default(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from pix2latent_amd import lanes

    class Ok(object):
        lanes_ok = True
    monkeypatch.delenv('P2L_SUBLANES', raising=False)
    assert not lanes.sub_wanted(3, Ok(), Ok()) and not lanes.sub_wanted(6, Ok(), Ok())
    assert lanes.sub_wanted(7, Ok(), Ok()) and lanes.sub_wanted(9, Ok(), Ok())
    monkeypatch.setenv('P2L_STREAMS', '1')
    assert not lanes.sub_wanted(9, Ok(), Ok())
