"""Generate tests/golden/*.npz by IMPORTING the reference (read-only, from
/root/reference) with in-memory stubs for the third-party packages that are
absent in this container.  Only the resulting numbers are committed; no
reference source travels.  Run:  python tools/make_golden.py

Stubs: easydict (15-line attr dict), cv2 / torchvision / lpips / nevergrad /
pytorch_pretrained_biggan (empty modules), `cma` (a recording fake whose ask()
returns seeded arrays) and torch.Tensor.cuda = identity (the reference
hard-codes .cuda(), variable_manager.py:217).
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(EasyDict(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x for x in v)
        super().__setattr__(k, v)
        super().__setitem__(k, v)
    __setitem__ = __setattr__


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    mod('easydict', EasyDict=EasyDict)
    mod('cv2')
    tv = mod('torchvision')
    tv.transforms = mod('torchvision.transforms')
    tv.transforms.functional = mod('torchvision.transforms.functional')

    def make_grid(x, nrow=8, padding=2, pad_value=0):
        return x  # the collage is not pinned
    tv.utils = mod('torchvision.utils', make_grid=make_grid)
    mod('lpips')
    from _toy import fake_nevergrad
    _ng = fake_nevergrad()
    mod('nevergrad', optimizers=_ng.optimizers, p=_ng.p)
    mod('pytorch_pretrained_biggan')
    from _toy import FakeCMAES
    mod('cma', CMAEvolutionStrategy=FakeCMAES)
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def compose_section():
    """(7) ComposeTransform (reference transform/transform_utils.py:122-184): two spatial
    transformations with different defaults / weights driven by one 6-vector"""
    import warnings
    warnings.simplefilter('ignore')
    from pix2latent.transform import SpatialTransform
    from pix2latent.transform.transform_utils import ComposeTransform
    g = torch.Generator().manual_seed(71)
    ims = torch.rand(3, 3, 16, 16, generator=g) * 2 - 1
    a, b = SpatialTransform(), SpatialTransform(t=[0.9, 0.05, -0.1], sensitivity=0.2)
    comp = ComposeTransform([(a, 1.0), (b, 2.0)])
    t = torch.tensor([[0.3, -0.5, 1.0, 0.8, 0.2, -0.4]])
    tb = t.repeat(3, 1) * torch.tensor([[1.0], [0.5], [-1.0]])
    single = ComposeTransform([SpatialTransform()])
    np.savez(os.path.join(OUT, 'compose_transform.npz'), ims=ims.numpy(), t=t.numpy(), tb=tb.numpy(),
             param=np.stack(comp.get_param()), param_flat=comp.get_param(as_tensor=True).numpy(),
             fwd=comp(ims, t).numpy(), fwd_b=comp(ims, tb).numpy(),
             inv_b=comp(comp(ims, tb), tb, invert=True).numpy(),
             spatial_only=comp(ims, tb, only_spatial=True).numpy(),
             single=single(ims, tb[:, :3]).numpy(),
             reweight=comp.reweight(tb[:, :3], 2.0, torch.tensor([1.0, 0.0, 0.0])).numpy())
    print('compose_transform.npz written')


def main():
    install_stubs()
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ['compose']:
        import pix2latent  # noqa: F401  (the reference)
        assert pix2latent.__file__.startswith(REF)
        return compose_section()
    import pix2latent  # noqa: F401  (the reference)
    from pix2latent import VariableManager, distribution
    from pix2latent.variable_manager import split_vars
    from pix2latent.utils import function_hooks as hook
    from pix2latent.utils.image import binarize
    import pix2latent.loss_functions as LF
    from pix2latent.optimizer.gradient_optimizer import GradientOptimizer
    from pix2latent.optimizer.basincma_optimizer import BasinCMAOptimizer
    from pix2latent.optimizer.cma_optimizer import CMAOptimizer
    from pix2latent.optimizer.closure import step as ref_step
    from _toy import ToyGenerator, toy_target, toy_weight, FakeCMAES
    assert pix2latent.__file__.startswith(REF)

    # (1) distribution
    torch.manual_seed(11)
    d1 = distribution.TruncatedNormalModulo(sigma=3.0, trunc=1.0)(5, (8,))
    torch.manual_seed(12)
    d2 = distribution.normal(0.5)(4, (3,))
    np.savez(os.path.join(OUT, 'distribution.npz'), tnm=d1.numpy(), normal=d2.numpy())

    # (2) losses
    g = torch.Generator().manual_seed(21)
    o = torch.rand(3, 3, 5, 5, generator=g) * 2 - 1
    t = torch.rand(3, 3, 5, 5, generator=g) * 2 - 1
    w = torch.rand(3, 3, 5, 5, generator=g)
    m = (torch.rand(3, 3, 5, 5, generator=g) > 0.4).float()
    per_map = torch.rand(3, 1, 5, 5, generator=g)
    rl1, rl2 = LF.ReconstructionLoss('l1'), LF.ReconstructionLoss('l2')
    _w = m * w
    per_weighted = torch.sum(per_map * _w, [1, 2, 3]) / torch.sum(_w, [1, 2, 3])   # loss_functions.py:143-147 body
    np.savez(os.path.join(OUT, 'losses.npz'), o=o.numpy(), t=t.numpy(), w=w.numpy(), m=m.numpy(),
             per_map=per_map.numpy(),
             l1=LF.l1_loss(o, t).numpy(), l2=LF.l2_loss(o, t).numpy(),
             masked_l1=LF.masked_l1_loss(o, t[:1], m[:1]).numpy(),
             masked_l2=LF.masked_l2_loss(o, t, m).numpy(),
             rec_l1_w=rl1(o, t, w).numpy(), rec_l1_wm=rl1(o, t, w, m).numpy(),
             rec_l2_w=rl2(o, t, w).numpy(), rec_l1_none=rl1(o, t).numpy(),
             per_weighted=per_weighted.numpy(),
             binarize=binarize((w * 0 + (w > 0.5).float() * 0.9995 + 0.0004).clone()).numpy())

    # (3) hooks
    g = torch.Generator().manual_seed(31)
    vs = [torch.randn(6, generator=g) * 3 for _ in range(3)]
    c = [v.clone() for v in vs]
    hook.Clamp(2.0)(c)
    n = [v.clone() for v in vs]
    hook.Normalize()(n)
    torch.manual_seed(32)
    p = [v.clone() for v in vs]
    hook.NormalPerturb(0.05)(p)
    torch.manual_seed(33)
    cp = [v.clone() for v in vs]
    hook.Compose(hook.NormalPerturb(0.05), hook.Clamp(2.0))(cp)
    np.savez(os.path.join(OUT, 'hooks.npz'), src=torch.stack(vs).numpy(), clamp=torch.stack(c).numpy(),
             normalize=torch.stack(n).numpy(), perturb=torch.stack(p).numpy(),
             compose=torch.stack(cp).numpy())

    # (4) VariableManager + split_vars + closure.step / GradientOptimizer traces
    def make_vm():
        vm = VariableManager()
        vm.register('z', (6,), 'input', distribution=distribution.TruncatedNormalModulo(),
                    learning_rate=0.05, hook_fn=hook.Clamp(1.5), grad_free=True)
        vm.register('c', (4,), 'input', default=torch.linspace(-0.2, 0.2, 4), learning_rate=0.01)
        vm.register('target', (3, 4, 4), 'output', requires_grad=False, default=toy_target())
        vm.register('weight', (3, 4, 4), 'output', requires_grad=False, default=toy_weight())
        return vm

    def toy_loss(out, target, weight):
        return LF.ReconstructionLoss()(out, target, weight)

    torch.manual_seed(41)
    vm = make_vm()
    v = vm.initialize(5)
    chunks = split_vars(v, 2)
    split = dict(
        n_chunks=len(chunks), sizes=np.array([c_.num_samples for c_ in chunks]),
        n_groups=len(v.opt.param_groups),
        group_lr=np.array([g_['lr'] for g_ in v.opt.param_groups]),
        z_init=torch.stack(v.input.z.data).detach().numpy(),
        c_init=torch.stack(v.input.c.data).detach().numpy(),
        z_requires_grad=np.array([t_.requires_grad for t_ in v.input.z.data]),
        target_requires_grad=np.array([t_.requires_grad for t_ in v.output.target.data]))
    for (N_, S_) in [(18, 9), (32, 9), (5, 2)]:
        vv = vm.initialize(N_)
        split['sizes_%d_%d' % (N_, S_)] = np.array([c_.num_samples for c_ in split_vars(vv, S_)])
    np.savez(os.path.join(OUT, 'variable_manager.npz'), **split)

    model = ToyGenerator()
    torch.manual_seed(42)
    opt = GradientOptimizer(model, make_vm(), toy_loss, max_batch_size=2)
    variables, outs, losses = opt.optimize(num_samples=5, grad_steps=3)
    tracked_z = torch.stack(opt.tracked['z']).numpy()
    # per-step losses via a second run that records after each step
    model2 = ToyGenerator()
    torch.manual_seed(42)
    opt2 = GradientOptimizer(model2, make_vm(), toy_loss, max_batch_size=2)
    vars2 = opt2.var_manager.initialize(num_samples=5)
    step_losses, step_z = [], []
    for i in range(3):
        _, l, _ = opt2.step(vars2, optimize=True, transform=(i == 0))
        step_losses.append(np.array(l))
        step_z.append(torch.stack(vars2.input.z.data).detach().numpy().copy())
    # a re-score (optimize=False) must not move anything but still runs hooks + forward
    z_before = torch.stack(vars2.input.z.data).detach().clone()
    out_ns, l_ns, _ = opt2.step(vars2, optimize=False)
    z_after = torch.stack(vars2.input.z.data).detach().clone()
    adam_steps = np.array([int(vars2.opt.state[p_]['step']) for g_ in vars2.opt.param_groups for p_ in g_['params']])
    np.savez(os.path.join(OUT, 'gradient_optimizer.npz'),
             final_z=torch.stack(variables.input.z.data).detach().numpy(),
             final_c=torch.stack(variables.input.c.data).detach().numpy(),
             final_loss=np.array(losses[-1][1]['loss']), n_steps=losses[-1][0],
             tracked_z=tracked_z, model_calls=np.array(model.calls),
             step_losses=np.stack(step_losses), step_z=np.stack(step_z),
             rescore_loss=np.array(l_ns), rescore_out=out_ns.detach().numpy(),
             rescore_dz=(z_after - z_before).abs().max().item(), adam_steps=adam_steps,
             rescore_model_calls=np.array(model2.calls[-3:]))

    # (5) BasinCMA / CMA control flow with the recording fake cma
    FakeCMAES.log = []
    model3 = ToyGenerator()
    torch.manual_seed(43)
    bopt = BasinCMAOptimizer(model3, make_vm(), toy_loss, max_batch_size=3)
    bvars, bouts, blosses = bopt.optimize(meta_steps=2, grad_steps=2, last_grad_steps=3)
    told = FakeCMAES.log
    np.savez(os.path.join(OUT, 'basincma.npz'),
             popsize=bopt.num_samples, n_tell=len(told),
             tell_x0=told[0][0], tell_y0=told[0][1], tell_x1=told[1][0], tell_y1=told[1][1],
             final_z=torch.stack(bvars.input.z.data).detach().numpy(),
             final_c=torch.stack(bvars.input.c.data).detach().numpy(),
             final_loss=np.array(blosses[-1][1]['loss']), total_steps=blosses[-1][0],
             model_calls=np.array(model3.calls))

    FakeCMAES.log = []
    model4 = ToyGenerator()
    torch.manual_seed(44)
    copt = CMAOptimizer(model4, make_vm(), toy_loss, max_batch_size=3)
    cvars, couts, closses = copt.optimize(meta_steps=3, grad_steps=2)
    told = FakeCMAES.log
    np.savez(os.path.join(OUT, 'cma.npz'), n_tell=len(told),
             tell_x2=told[2][0], tell_y2=told[2][1],
             final_z=torch.stack(cvars.input.z.data).detach().numpy(),
             final_loss=np.array(closses[-1][1]['loss']), total_steps=closses[-1][0],
             model_calls=np.array(model4.calls))
    # (5b) Nevergrad / HybridNevergrad control flow with the recording fake nevergrad
    from _toy import FakeNGOpt
    from pix2latent.optimizer.ng_optimizer import NevergradOptimizer
    from pix2latent.optimizer.hybrid_ng_optimizer import HybridNevergradOptimizer

    def ng_trace():
        kinds = np.array([0 if k == 'ask' else 1 for k, _ in FakeNGOpt.log])
        tells = [p_ for k, p_ in FakeNGOpt.log if k == 'tell']
        return dict(kinds=kinds, tell_uid=np.array([t_[0] for t_ in tells]),
                    tell_x=np.stack([t_[1] for t_ in tells]),
                    tell_y=np.array([t_[2] for t_ in tells]),
                    budget=FakeNGOpt.instances[-1].budget)

    FakeNGOpt.log, FakeNGOpt.instances = [], []
    model5 = ToyGenerator()
    torch.manual_seed(45)
    nopt = NevergradOptimizer('CMA', model5, make_vm(), toy_loss, max_batch_size=3)
    nvars, nouts, nlosses = nopt.optimize(num_samples=4, meta_steps=3, grad_steps=2)
    np.savez(os.path.join(OUT, 'nevergrad.npz'),
             final_z=torch.stack(nvars.input.z.data).detach().numpy(),
             final_c=torch.stack(nvars.input.c.data).detach().numpy(),
             final_loss=np.array(nlosses[-1][1]['loss']), total_steps=nlosses[-1][0],
             model_calls=np.array(model5.calls), **ng_trace())

    FakeNGOpt.log, FakeNGOpt.instances = [], []
    model6 = ToyGenerator()
    torch.manual_seed(46)
    hopt = HybridNevergradOptimizer('CMA', model6, make_vm(), toy_loss, max_batch_size=3)
    hvars, houts, hlosses = hopt.optimize(num_samples=4, meta_steps=2, grad_steps=2, last_grad_steps=3)
    np.savez(os.path.join(OUT, 'hybrid_nevergrad.npz'),
             final_z=torch.stack(hvars.input.z.data).detach().numpy(),
             final_c=torch.stack(hvars.input.c.data).detach().numpy(),
             final_loss=np.array(hlosses[-1][1]['loss']), total_steps=hlosses[-1][0],
             model_calls=np.array(model6.calls), **ng_trace())

    # (6) SpatialTransform / pre-alignment / TransformBasinCMAOptimizer
    import warnings
    warnings.simplefilter('ignore')
    from pix2latent.transform import SpatialTransform, TransformBasinCMAOptimizer
    from pix2latent.transform.transform_utils import (compute_pre_alignment, bbox_from_mask,
                                                      compute_stat_from_mask, convert_to_t)
    g = torch.Generator().manual_seed(61)
    ims = torch.rand(3, 3, 16, 16, generator=g) * 2 - 1
    tpar = torch.tensor([[1.0, 0.0, 0.0], [0.8, 0.1, -0.2], [1.3, -0.25, 0.15]])
    st = SpatialTransform()
    fwd = st.transform(ims, tpar)
    inv = st.invert_transform(fwd, tpar)
    st2 = SpatialTransform(t=[0.9, 0.05, -0.1], sensitivity=0.1)
    delta = torch.tensor([[0.5, -1.0, 2.0]]).repeat(3, 1)
    called = st2(ims, delta)
    called_inv = st2(ims, delta, invert=True)
    mask = torch.zeros(3, 32, 32)
    mask[:, 6:20, 10:29] = 1.0
    pre = compute_pre_alignment(mask.clone())
    st3 = SpatialTransform(pre_align=mask.clone())
    np.savez(os.path.join(OUT, 'spatial_transform.npz'), ims=ims.numpy(), t=tpar.numpy(),
             fwd=fwd.numpy(), inv=inv.numpy(), delta=delta.numpy(), called=called.numpy(),
             called_inv=called_inv.numpy(), mask=mask.numpy(), pre_align=np.asarray(pre),
             bbox=np.array(bbox_from_mask(mask)),
             stat=np.array(compute_stat_from_mask(mask)).reshape(-1),
             convert=convert_to_t((0.4, 0.6), (0.5, 0.3), (0.5, 0.5), (0.8, 0.8)).numpy(),
             default_param=st3.get_default_param().numpy())

    def make_vm_t():
        vm = VariableManager()
        vm.register('z', (6,), 'input', distribution=distribution.TruncatedNormalModulo(),
                    learning_rate=0.05, hook_fn=hook.Clamp(1.5))
        vm.register('c', (4,), 'input', default=torch.linspace(-0.2, 0.2, 4), learning_rate=0.01)
        vm.register('target', (3, 4, 4), 'output', requires_grad=False, default=toy_target())
        vm.register('weight', (3, 4, 4), 'output', requires_grad=False, default=toy_weight())
        vm.register('t', (3,), 'transform', requires_grad=False, grad_free=True)
        return vm
    FakeCMAES.log = []
    model5 = ToyGenerator()
    torch.manual_seed(45)
    topt = TransformBasinCMAOptimizer(model5, make_vm_t(), toy_loss, max_batch_size=4)
    topt.register_transform(SpatialTransform(), 't', 'target')
    topt.register_transform(SpatialTransform(), 't', 'weight')
    topt.set_variable_propagation('z')
    tvars, (tout, ttarget, tcand), tloss = topt.optimize(meta_steps=3, grad_steps=2)
    told = FakeCMAES.log
    np.savez(os.path.join(OUT, 'transform_basincma.npz'), popsize=topt.num_samples,
             n_tell=len(told), tell_x0=told[0][0], tell_y0=told[0][1], tell_y1=told[1][1],
             tracked=torch.stack(topt.transform_tracked).numpy(),
             candidate=topt.get_candidate().numpy(), best_loss=float(topt._best_loss),
             final_z=torch.stack(tvars.input.z.data).detach().numpy(),
             final_target=torch.stack(tvars.output.target.data).detach().numpy(),
             final_loss=np.array(tloss), cand_out=tcand.detach().numpy(),
             vp_mean=topt.vp_means['z'].detach().numpy(), model_calls=np.array(model5.calls))
    compose_section()
    print('golden fixtures written to', OUT)
    for f in sorted(os.listdir(OUT)):
        print('  ', f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
