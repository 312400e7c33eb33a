"""SpatialTransform / pre-alignment / TransformBasinCMAOptimizer against golden
vectors captured from the imported reference (tools/make_golden.py). CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _toy import ToyGenerator, toy_target, toy_weight, FakeCMAES  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def gold(name):
    return np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False)


def test_spatial_transform_golden():
    from pix2latent_amd.transform import SpatialTransform
    g = gold('spatial_transform')
    ims, t = torch.from_numpy(g['ims']), torch.from_numpy(g['t'])
    st = SpatialTransform()
    fwd = st.transform(ims, t)
    assert np.allclose(fwd.numpy(), g['fwd'], atol=1e-6)
    assert np.allclose(st.invert_transform(fwd, t).numpy(), g['inv'], atol=1e-6)
    assert np.allclose(fwd[0].numpy(), g['ims'][0], atol=1e-6)          # t = identity
    st2 = SpatialTransform(t=[0.9, 0.05, -0.1], sensitivity=0.1)
    delta = torch.from_numpy(g['delta'])
    assert np.allclose(st2(ims, delta).numpy(), g['called'], atol=1e-6)
    assert np.allclose(st2(ims, delta, invert=True).numpy(), g['called_inv'], atol=1e-6)


def test_pre_alignment_golden():
    from pix2latent_amd.transform import SpatialTransform
    from pix2latent_amd.transform.transform_utils import (compute_pre_alignment, bbox_from_mask,
                                                          compute_stat_from_mask, convert_to_t)
    g = gold('spatial_transform')
    mask = torch.from_numpy(g['mask'])
    assert list(bbox_from_mask(mask)) == list(g['bbox'])
    assert np.allclose(np.array(compute_stat_from_mask(mask)).reshape(-1), g['stat'])
    assert np.allclose(convert_to_t((0.4, 0.6), (0.5, 0.3), (0.5, 0.5), (0.8, 0.8)).numpy(), g['convert'])
    assert np.allclose(compute_pre_alignment(mask.clone()), g['pre_align'])
    assert np.allclose(SpatialTransform(pre_align=mask.clone()).get_default_param().numpy(),
                       g['default_param'])
    assert list(bbox_from_mask(torch.zeros(3, 8, 8))) == [0, 0, 8, 8]     # empty mask: full range


def test_transform_basincma_control_flow_golden(monkeypatch):
    from pix2latent_amd import VariableManager, distribution
    from pix2latent_amd.utils import function_hooks as hook
    from pix2latent_amd.transform import SpatialTransform, TransformBasinCMAOptimizer
    import pix2latent_amd.optimizer.base_cma_optimizer as B
    from oracle.lpips_ref import reconstruction_loss
    g = gold('transform_basincma')
    FakeCMAES.log = []
    monkeypatch.setattr(B, 'CMAEvolutionStrategy', FakeCMAES)
    vm = VariableManager(device='cpu')
    vm.register('z', (6,), 'input', distribution=distribution.TruncatedNormalModulo(),
                learning_rate=0.05, hook_fn=hook.Clamp(1.5))
    vm.register('c', (4,), 'input', default=torch.linspace(-0.2, 0.2, 4), learning_rate=0.01)
    vm.register('target', (3, 4, 4), 'output', requires_grad=False, default=toy_target())
    vm.register('weight', (3, 4, 4), 'output', requires_grad=False, default=toy_weight())
    vm.register('t', (3,), 'transform', requires_grad=False, grad_free=True)
    model = ToyGenerator()
    torch.manual_seed(45)
    topt = TransformBasinCMAOptimizer(model, vm, lambda o, target, weight: reconstruction_loss(o, target, weight),
                                      max_batch_size=4)
    topt.register_transform(SpatialTransform(), 't', 'target')
    topt.register_transform(SpatialTransform(), 't', 'weight')
    topt.set_variable_propagation('z')
    tvars, (tout, ttarget, tcand), tloss = topt.optimize(meta_steps=3, grad_steps=2)
    told = FakeCMAES.log
    assert topt.num_samples == int(g['popsize']) == 7
    assert len(told) == int(g['n_tell'])
    assert np.allclose(told[0][0], g['tell_x0'], atol=1e-12)
    assert np.allclose(told[0][1], g['tell_y0'], atol=1e-6)
    assert np.allclose(told[1][1], g['tell_y1'], atol=1e-6)
    assert np.allclose(torch.stack(topt.transform_tracked).numpy(), g['tracked'], atol=1e-6)
    assert np.allclose(topt.get_candidate().numpy(), g['candidate'], atol=1e-6)
    assert abs(float(topt._best_loss) - float(g['best_loss'])) < 1e-6
    assert np.allclose(torch.stack(list(tvars.input.z.data)).detach().numpy(), g['final_z'], atol=1e-5)
    assert np.allclose(torch.stack(list(tvars.output.target.data)).numpy(), g['final_target'], atol=1e-6)
    assert np.allclose(np.array(tloss), g['final_loss'], atol=1e-6)
    assert np.allclose(tcand.numpy(), g['cand_out'], atol=1e-6)
    assert np.allclose(topt.vp_means['z'].numpy(), g['vp_mean'], atol=1e-5)
    assert [c[0] for c in model.calls] == [int(c[0]) for c in g['model_calls']]


def test_compose_transform_golden():
    """ComposeTransform vs the imported reference (transform_utils.py:122-184): slicing of
    the shared parameter vector, re-weighting around each default, t broadcast over the
    batch, inversion order, only_spatial."""
    from pix2latent_amd.transform import SpatialTransform, ComposeTransform
    g = gold('compose_transform')
    ims, t, tb = (torch.from_numpy(g[k]) for k in ('ims', 't', 'tb'))
    comp = ComposeTransform([(SpatialTransform(), 1.0),
                             (SpatialTransform(t=[0.9, 0.05, -0.1], sensitivity=0.2), 2.0)])
    assert np.allclose(np.stack(comp.get_param()), g['param'])
    assert np.allclose(comp.get_param(as_tensor=True).numpy(), g['param_flat'])
    assert np.allclose(comp(ims, t).numpy(), g['fwd'], atol=1e-6)
    assert np.allclose(comp(ims, tb).numpy(), g['fwd_b'], atol=1e-6)
    assert np.allclose(comp(comp(ims, tb), tb, invert=True).numpy(), g['inv_b'], atol=1e-6)
    assert np.allclose(comp(ims, tb, only_spatial=True).numpy(), g['spatial_only'], atol=1e-6)
    assert np.allclose(ComposeTransform([SpatialTransform()])(ims, tb[:, :3]).numpy(), g['single'], atol=1e-6)
    assert np.allclose(comp.reweight(tb[:, :3], 2.0, torch.tensor([1.0, 0.0, 0.0])).numpy(), g['reweight'])
    assert comp.get_opt_param().shape == (6,)
    assert 'ComposeTransform' in str(comp)


def test_setup_transform_fn():
    """setup_transform_fn (transform_utils.py:15-50): nothing requested -> (None, None);
    spatial search -> one SpatialTransform with the identity start; align -> the start is the
    mask's pre-alignment (golden value); colour names are refused (out of scope)."""
    import types
    from pix2latent_amd.transform import setup_transform_fn, ComposeTransform
    g = gold('spatial_transform')
    mask = torch.from_numpy(g['mask'])
    ns = types.SimpleNamespace
    assert setup_transform_fn(ns(spatial_transform=False, align=False, color_transform=[]), mask) == (None, None)
    fn, t = setup_transform_fn(ns(spatial_transform=True, align=False, color_transform=[]), mask)
    assert isinstance(fn, ComposeTransform) and t.shape == (1, 3)
    assert np.allclose(t.numpy(), [[1.0, 0.0, 0.0]])
    fn, t = setup_transform_fn(ns(spatial_transform=False, align=True, color_transform=[]), mask.clone())
    assert np.allclose(t.numpy()[0], g['pre_align'])
    with pytest.raises(NotImplementedError):
        setup_transform_fn(ns(spatial_transform=True, align=False, color_transform=['hue']), mask)


def test_spatial_transform_checks_batch_and_is_differentiable():
    from pix2latent_amd.transform import SpatialTransform
    st = SpatialTransform()
    ims = torch.rand(2, 3, 8, 8)
    with pytest.raises(AssertionError, match='one transformation per image'):
        st.transform(ims, torch.tensor([[1.0, 0.0, 0.0]] * 3))
    t = torch.tensor([[0.9, 0.1, 0.0], [1.1, 0.0, -0.1]], requires_grad=True)
    st.transform(ims, t).sum().backward()
    assert t.grad is not None and t.grad.abs().sum() > 0
