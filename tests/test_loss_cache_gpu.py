"""The target-feature cache of the native losses (pix2latent_amd/loss_functions._LossEngine): a CMA
generation starts from fresh variables (/root/reference pix2latent/optimizer/base_cma_optimizer.py:79) whose
targets have the old CONTENT; the LPIPS network must not run over them again -- and must run when the content
differs or when the tensors a slot was prepared from were modified in place."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_equal_content_reuses_the_prepared_target_features(dev):
    import pix2latent_amd.loss_functions as LF
    from pix2latent_amd.utils import synthetic as S
    loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=S.lpips_vgg_weights(1), device=dev)
    eng = loss_fn._engine
    calls = []
    real = eng.f_prepare
    eng.f_prepare = lambda *a: (calls.append(1), real(*a))[1]
    g = torch.Generator().manual_seed(0)
    B, H = 3, 64
    out = torch.rand(B, 3, H, H, generator=g).to(dev) * 2 - 1
    tgt = (torch.rand(B, 3, H, H, generator=g) * 2 - 1).to(dev)
    wgt = torch.rand(B, 3, H, H, generator=g).to(dev)
    l0 = loss_fn(out, tgt, wgt).clone()
    assert len(calls) == 1
    assert torch.equal(loss_fn(out, tgt, wgt), l0) and len(calls) == 1            # same tensors: the key
    t2, w2 = tgt.clone(), wgt.clone()                                              # fresh tensors, equal content
    assert torch.equal(loss_fn(out, t2, w2), l0) and len(calls) == 1
    assert torch.equal(loss_fn(out, tgt, wgt), l0) and len(calls) == 1            # both keys stay valid
    t3 = tgt.clone()
    t3[1, 0, 5, 7] += 0.25                                                         # one pixel differs
    l3 = loss_fn(out, t3, w2)
    assert len(calls) == 2 and not torch.equal(l3, l0) and torch.equal(l3[0], l0[0])
    # in-place modification of the tensors a slot was prepared from: their old content is gone
    tgt.mul_(0.5)
    l4 = loss_fn(out, tgt, wgt)
    assert len(calls) == 3
    fresh = LF.ProjectionLoss(lpips_net='vgg', weights=S.lpips_vgg_weights(1), device=dev)
    assert torch.equal(fresh(out, tgt.clone(), wgt.clone()), l4)
    # ... and a clone of the OLD content must not be served by that slot
    assert torch.equal(loss_fn(out, t2.clone(), w2.clone()), l0)
