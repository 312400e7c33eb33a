"""Host logic (VariableManager / closure / optimizers / hooks / losses /
distributions) pinned against golden vectors captured from the IMPORTED
reference (tools/make_golden.py -> tests/golden/*.npz).  CPU only."""
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


def make_vm():
    from pix2latent_amd import VariableManager, distribution
    from pix2latent_amd.utils import function_hooks as hook
    vm = VariableManager(device='cpu')
    vm.register('z', (6,), 'input', distribution=distribution.TruncatedNormalModulo(),
                learning_rate=0.05, hook_fn=hook.Clamp(1.5), grad_free=True)
    vm.register('c', (4,), 'input', default=torch.linspace(-0.2, 0.2, 4), learning_rate=0.01)
    vm.register('target', (3, 4, 4), 'output', requires_grad=False, default=toy_target())
    vm.register('weight', (3, 4, 4), 'output', requires_grad=False, default=toy_weight())
    return vm


def toy_loss(out, target, weight):
    from oracle.lpips_ref import reconstruction_loss
    return reconstruction_loss(out, target, weight)


def test_distribution_golden():
    from pix2latent_amd import distribution
    g = gold('distribution')
    torch.manual_seed(11)
    d1 = distribution.TruncatedNormalModulo(sigma=3.0, trunc=1.0)(5, (8,))
    torch.manual_seed(12)
    d2 = distribution.normal(0.5)(4, (3,))
    assert np.array_equal(d1.numpy(), g['tnm'])      # sigma/trunc args ignored, as in the reference
    assert np.array_equal(d2.numpy(), g['normal'])


def test_losses_golden_oracle_and_host():
    """oracle restatement AND the package's torch-op losses vs the reference's numbers"""
    from oracle import lpips_ref as L
    import importlib
    g = gold('losses')
    o, t, w, m = (torch.from_numpy(g[k]) for k in ('o', 't', 'w', 'm'))
    assert np.array_equal(L.l1_loss(o, t).numpy(), g['l1'])
    assert np.array_equal(L.l2_loss(o, t).numpy(), g['l2'])
    assert np.allclose(L.masked_l1_loss(o, t[:1], m[:1]).numpy(), g['masked_l1'], atol=0, rtol=1e-6)
    assert np.allclose(L.masked_l2_loss(o, t, m).numpy(), g['masked_l2'], atol=0, rtol=1e-6)
    assert np.array_equal(L.reconstruction_loss(o, t, w).numpy(), g['rec_l1_w'])
    assert np.array_equal(L.reconstruction_loss(o, t, w, m).numpy(), g['rec_l1_wm'])
    assert np.array_equal(L.reconstruction_loss(o, t, w, loss_type='l2').numpy(), g['rec_l2_w'])
    assert np.array_equal(L.reconstruction_loss(o, t).numpy(), g['rec_l1_none'])
    pm = torch.from_numpy(g['per_map'])
    assert np.array_equal(L._weighted(pm, w, m).numpy(), g['per_weighted'])
    try:
        LF = importlib.import_module('pix2latent_amd.loss_functions')
    except Exception:
        pytest.skip('native library not built')
    assert np.array_equal(LF.l1_loss(o, t).numpy(), g['l1'])
    assert np.allclose(LF.masked_l1_loss(o, t[:1], m[:1]).numpy(), g['masked_l1'], rtol=1e-6)
    assert np.array_equal(LF.ReconstructionLoss('l2')(o, t, w).numpy(), g['rec_l2_w'])
    assert np.array_equal(LF.ReconstructionLoss()(o, t, w, m).numpy(), g['rec_l1_wm'])   # CPU tensors: torch ops


def test_binarize_golden():
    from pix2latent_amd.utils.image import binarize
    g = gold('losses')
    w = torch.from_numpy(g['w'])
    x = (w * 0 + (w > 0.5).float() * 0.9995 + 0.0004).clone()
    assert np.array_equal(binarize(x).numpy(), g['binarize'])


def test_hooks_golden():
    from pix2latent_amd.utils import function_hooks as hook
    g = gold('hooks')
    src = torch.from_numpy(g['src'])

    def fresh():
        return [v.clone() for v in src]
    c = fresh(); hook.Clamp(2.0)(c)
    assert np.array_equal(torch.stack(c).numpy(), g['clamp'])
    n = fresh(); hook.Normalize()(n)
    assert np.allclose(torch.stack(n).numpy(), g['normalize'], rtol=1e-6)
    torch.manual_seed(32)
    p = fresh(); hook.NormalPerturb(0.05)(p)
    assert np.array_equal(torch.stack(p).numpy(), g['perturb'])
    torch.manual_seed(33)
    cp = fresh(); hook.Compose(hook.NormalPerturb(0.05), hook.Clamp(2.0))(cp)
    assert np.array_equal(torch.stack(cp).numpy(), g['compose'])
    # batched forms (one kernel per chunk) agree with the per-sample forms
    b = src.clone(); hook.Clamp(2.0).apply_batched(b)
    assert np.array_equal(b.numpy(), g['clamp'])
    b = src.clone(); hook.Normalize().apply_batched(b)
    assert np.allclose(b.numpy(), g['normalize'], rtol=1e-5)


def test_variable_manager_and_split_golden():
    from pix2latent_amd.variable_manager import split_vars
    g = gold('variable_manager')
    torch.manual_seed(41)
    vm = make_vm()
    v = vm.initialize(5)
    chunks = split_vars(v, 2)
    assert len(chunks) == int(g['n_chunks'])
    assert [c.num_samples for c in chunks] == list(g['sizes'])
    assert len(v.opt.param_groups) == int(g['n_groups'])
    assert np.allclose([pg['lr'] for pg in v.opt.param_groups], g['group_lr'])
    assert np.array_equal(torch.stack(list(v.input.z.data)).detach().numpy(), g['z_init'])
    assert np.array_equal(torch.stack(list(v.input.c.data)).detach().numpy(), g['c_init'])
    assert [t.requires_grad for t in v.input.z.data] == list(g['z_requires_grad'])
    assert [t.requires_grad for t in v.output.target.data] == list(g['target_requires_grad'])
    for (N_, S_) in [(18, 9), (32, 9), (5, 2)]:
        vv = vm.initialize(N_)
        assert [c.num_samples for c in split_vars(vv, S_)] == list(g['sizes_%d_%d' % (N_, S_)])
    # per-sample tensors are views of one contiguous buffer
    assert v.input.z.data[3].data_ptr() == v.input.z.buf[3].data_ptr()
    v.input.z.data[3].data.clamp_(-0.1, 0.1)
    assert v.input.z.buf[3].abs().max() <= 0.1


def test_register_asserts_like_reference():
    from pix2latent_amd import VariableManager
    vm = VariableManager(device='cpu')
    with pytest.raises(AssertionError, match='default and shape must match'):
        vm.register('c', (4,), 'input', default=torch.zeros(5))
    assert vm.register('z', (4,), 'input') is True
    assert vm.register('z', (4,), 'input') is False          # duplicate
    assert vm.edit_variable('nope', {}) is False
    assert vm.edit_variable('z', {'learning_rate': 0.1}) is True
    assert vm.variable_info['z']['learning_rate'] == 0.1


def test_gradient_optimizer_trajectory_golden():
    """3 Adam steps, 5 samples, chunks 2+2+1: losses, latents, call pattern."""
    from pix2latent_amd.optimizer import GradientOptimizer
    g = gold('gradient_optimizer')
    model = ToyGenerator()
    torch.manual_seed(42)
    opt = GradientOptimizer(model, make_vm(), toy_loss, max_batch_size=2)
    variables, outs, losses = opt.optimize(num_samples=5, grad_steps=3)
    assert losses[-1][0] == int(g['n_steps'])
    assert np.allclose(np.array(losses[-1][1]['loss']), g['final_loss'], rtol=0, atol=1e-7)
    assert np.allclose(torch.stack(list(variables.input.z.data)).detach().numpy(), g['final_z'], atol=1e-7)
    assert np.allclose(torch.stack(list(variables.input.c.data)).detach().numpy(), g['final_c'], atol=1e-7)
    assert np.allclose(torch.stack(opt.tracked['z']).numpy(), g['tracked_z'], atol=1e-7)
    assert [c[0] for c in model.calls] == [int(c[0]) for c in g['model_calls']]

    model2 = ToyGenerator()
    torch.manual_seed(42)
    opt2 = GradientOptimizer(model2, make_vm(), toy_loss, max_batch_size=2)
    vars2 = opt2.var_manager.initialize(num_samples=5)
    for i in range(3):
        _, l, _ = opt2.step(vars2, optimize=True, transform=(i == 0))
        assert np.allclose(np.array(l), g['step_losses'][i], atol=1e-7)
        assert np.allclose(torch.stack(list(vars2.input.z.data)).detach().numpy(), g['step_z'][i], atol=1e-7)
    z_before = torch.stack(list(vars2.input.z.data)).detach().clone()
    out_ns, l_ns, _ = opt2.step(vars2, optimize=False)
    assert np.allclose(np.array(l_ns), g['rescore_loss'], atol=1e-7)
    assert np.allclose(out_ns.detach().numpy(), g['rescore_out'], atol=1e-7)
    # a re-score still runs the hooks (Clamp pulls back what the last Adam step pushed out)
    dz = (torch.stack(list(vars2.input.z.data)).detach() - z_before).abs().max().item()
    assert abs(dz - float(g['rescore_dz'])) < 1e-7
    steps = [int(vars2.opt.state[p]['step']) for pg in vars2.opt.param_groups for p in pg['params']]
    assert steps == list(g['adam_steps'])
    assert [c[0] for c in model2.calls[-3:]] == [int(c[0]) for c in g['rescore_model_calls']]


@pytest.mark.parametrize('ebs', [5, 'all'])
def test_exec_batch_size_keeps_reference_trajectory(ebs):
    """executing the whole population in one pass (exec_batch_size) with the
    reference chunk's gradient scale reproduces the chunked golden trajectory."""
    from pix2latent_amd.optimizer import GradientOptimizer
    g = gold('gradient_optimizer')
    model = ToyGenerator()
    torch.manual_seed(42)
    opt = GradientOptimizer(model, make_vm(), toy_loss, max_batch_size=2, exec_batch_size=ebs)
    variables, outs, losses = opt.optimize(num_samples=5, grad_steps=3)
    assert [c[0] for c in model.calls] == [5, 5, 5]
    assert np.allclose(np.array(losses[-1][1]['loss']), g['final_loss'], atol=1e-6)
    assert np.allclose(torch.stack(list(variables.input.z.data)).detach().numpy(), g['final_z'], atol=1e-6)
    assert np.allclose(torch.stack(list(variables.input.c.data)).detach().numpy(), g['final_c'], atol=1e-6)


def test_exec_batch_from_environment(monkeypatch):
    """P2L_EXEC_BATCH lets an unmodified example script run the population in one pass"""
    from pix2latent_amd.optimizer import GradientOptimizer
    g = gold('gradient_optimizer')
    monkeypatch.setenv('P2L_EXEC_BATCH', 'all')
    model = ToyGenerator()
    torch.manual_seed(42)
    opt = GradientOptimizer(model, make_vm(), toy_loss, max_batch_size=2)
    assert opt.exec_batch_size == 'all'
    variables, outs, losses = opt.optimize(num_samples=5, grad_steps=3)
    assert [c[0] for c in model.calls] == [5, 5, 5]
    assert np.allclose(np.array(losses[-1][1]['loss']), g['final_loss'], atol=1e-6)


def _patch_cma(monkeypatch):
    import pix2latent_amd.optimizer.base_cma_optimizer as B
    FakeCMAES.log = []
    monkeypatch.setattr(B, 'CMAEvolutionStrategy', FakeCMAES)


def test_basincma_control_flow_golden(monkeypatch):
    """same fake CMA as the golden run: what is asked, what is told (asked x with
    REFINED losses), fresh Adam per generation, final latents."""
    from pix2latent_amd.optimizer import BasinCMAOptimizer
    g = gold('basincma')
    _patch_cma(monkeypatch)
    model = ToyGenerator()
    torch.manual_seed(43)
    bopt = BasinCMAOptimizer(model, make_vm(), toy_loss, max_batch_size=3)
    bvars, bouts, blosses = bopt.optimize(meta_steps=2, grad_steps=2, last_grad_steps=3)
    told = FakeCMAES.log
    assert bopt.num_samples == int(g['popsize'])
    assert len(told) == int(g['n_tell'])
    for k in (0, 1):
        assert np.allclose(told[k][0], g['tell_x%d' % k], atol=1e-12)
        assert np.allclose(told[k][1], g['tell_y%d' % k], atol=1e-7)
        assert np.array_equal(np.argsort(told[k][1]), np.argsort(g['tell_y%d' % k]))
    assert np.allclose(torch.stack(list(bvars.input.z.data)).detach().numpy(), g['final_z'], atol=1e-6)
    assert np.allclose(torch.stack(list(bvars.input.c.data)).detach().numpy(), g['final_c'], atol=1e-6)
    assert np.allclose(np.array(blosses[-1][1]['loss']), g['final_loss'], atol=1e-6)
    assert blosses[-1][0] == int(g['total_steps'])
    assert [c[0] for c in model.calls] == [int(c[0]) for c in g['model_calls']]


def test_cma_control_flow_golden(monkeypatch):
    from pix2latent_amd.optimizer import CMAOptimizer
    g = gold('cma')
    _patch_cma(monkeypatch)
    model = ToyGenerator()
    torch.manual_seed(44)
    copt = CMAOptimizer(model, make_vm(), toy_loss, max_batch_size=3)
    cvars, couts, closses = copt.optimize(meta_steps=3, grad_steps=2)
    told = FakeCMAES.log
    assert len(told) == int(g['n_tell'])
    assert np.allclose(told[2][0], g['tell_x2'], atol=1e-12)
    assert np.allclose(told[2][1], g['tell_y2'], atol=1e-7)
    assert np.allclose(torch.stack(list(cvars.input.z.data)).detach().numpy(), g['final_z'], atol=1e-6)
    assert np.allclose(np.array(closses[-1][1]['loss']), g['final_loss'], atol=1e-6)
    assert closses[-1][0] == int(g['total_steps'])
    assert [c[0] for c in model.calls] == [int(c[0]) for c in g['model_calls']]


def _patch_ng(monkeypatch):
    from _toy import FakeNGOpt, fake_nevergrad
    import pix2latent_amd.optimizer.base_ng_optimizer as B
    FakeNGOpt.log, FakeNGOpt.instances = [], []
    monkeypatch.setattr(B, 'ng', fake_nevergrad())
    return FakeNGOpt


def _check_ng_trace(F, g, variables, losses, model):
    kinds = np.array([0 if k == 'ask' else 1 for k, _ in F.log])
    tells = [p for k, p in F.log if k == 'tell']
    assert np.array_equal(kinds, g['kinds']), 'ask / tell call order'
    assert np.array_equal(np.array([t[0] for t in tells]), g['tell_uid']), 'which candidate is told'
    assert np.allclose(np.stack([t[1] for t in tells]), g['tell_x'], atol=1e-12)
    assert np.allclose(np.array([t[2] for t in tells]), g['tell_y'], atol=1e-6)
    assert F.instances[-1].budget == int(g['budget'])
    assert np.allclose(torch.stack(list(variables.input.z.data)).detach().numpy(), g['final_z'], atol=1e-6)
    assert np.allclose(torch.stack(list(variables.input.c.data)).detach().numpy(), g['final_c'], atol=1e-6)
    assert np.allclose(np.array(losses[-1][1]['loss']), g['final_loss'], atol=1e-6)
    assert losses[-1][0] == int(g['total_steps'])
    assert [c[0] for c in model.calls] == [int(c[0]) for c in g['model_calls']]


def test_nevergrad_control_flow_golden(monkeypatch):
    """NevergradOptimizer (reference ng_optimizer.py:22-91) with the recording fake
    nevergrad of the golden run: num_samples asks per round, forward-only scoring, every
    asked candidate told its loss, budget = meta_steps, Adam fine-tuning of the last draw."""
    from pix2latent_amd.optimizer import NevergradOptimizer
    g = gold('nevergrad')
    F = _patch_ng(monkeypatch)
    model = ToyGenerator()
    torch.manual_seed(45)
    nopt = NevergradOptimizer('CMA', model, make_vm(), toy_loss, max_batch_size=3)
    nvars, nouts, nlosses = nopt.optimize(num_samples=4, meta_steps=3, grad_steps=2)
    _check_ng_trace(F, g, nvars, nlosses, model)


def test_hybrid_nevergrad_control_flow_golden(monkeypatch):
    """HybridNevergradOptimizer (reference hybrid_ng_optimizer.py:23-81, BASELINE config 4):
    ask -> grad_steps of Adam -> re-score -> tell the ASKED candidates the REFINED losses;
    budget = meta_steps * grad_steps; last draw refined for last_grad_steps."""
    from pix2latent_amd.optimizer import HybridNevergradOptimizer
    g = gold('hybrid_nevergrad')
    F = _patch_ng(monkeypatch)
    model = ToyGenerator()
    torch.manual_seed(46)
    hopt = HybridNevergradOptimizer('CMA', model, make_vm(), toy_loss, max_batch_size=3)
    hvars, houts, hlosses = hopt.optimize(num_samples=4, meta_steps=2, grad_steps=2, last_grad_steps=3)
    _check_ng_trace(F, g, hvars, hlosses, model)


def test_ng_compat_facade_conventions():
    """our nevergrad replacement keeps the API shape the reference relies on"""
    from pix2latent_amd.optimizer import ng_compat as ng
    for method in ('CMA', 'DE', 'OnePlusOne', 'RandomSearch'):
        opt = ng.optimizers.registry[method](parametrization=ng.p.Array(init=np.zeros(6)), budget=20, seed=3)
        asked = [opt.ask() for _ in range(5)]
        stacked = np.concatenate([x.args for x in asked])       # reference base_ng_optimizer.py:107
        assert stacked.shape == (5, 6) and np.isfinite(stacked).all()
        for c in asked:
            opt.tell(c, float(np.sum(c.args[0] ** 2)))
        assert opt.num_ask == 5 and opt.num_tell == 5


def test_random_hooks_row_form_is_the_reference_stream():
    """apply_batched draws one randn per row in population order: bit-identical to the
    reference's per-sample loop (golden `perturb` / `compose`), for any execution chunking"""
    from pix2latent_amd.utils import function_hooks as hook
    g = gold('hooks')
    src = torch.from_numpy(g['src'])
    n = src.size(0)
    torch.manual_seed(32)
    b = src.clone(); hook.NormalPerturb(0.05).apply_batched(b)
    assert np.array_equal(b.numpy(), g['perturb'])
    torch.manual_seed(33)
    b = src.clone(); hook.Compose(hook.NormalPerturb(0.05), hook.Clamp(2.0)).apply_batched(b)
    # (Compose applies NormalPerturb to ALL rows, then Clamp: same order as the reference)
    assert np.array_equal(b.numpy(), g['compose'])
    # a rank holding rows [1,3) of the chunk draws-and-discards the others: same numbers on
    # its rows, and the generator ends where a single process would
    torch.manual_seed(32)
    b = src.clone(); hook.NormalPerturb(0.05).apply_batched(b[1:3], hook.HookSpan(1, 3, 0, n))
    assert np.array_equal(b[1:3].numpy(), g['perturb'][1:3])
    assert np.array_equal(b[0].numpy(), g['src'][0])
    after = torch.rand(1)
    torch.manual_seed(32)
    hook.NormalPerturb(0.05).apply_batched(src.clone())
    assert torch.equal(after, torch.rand(1))
    # ... and so does a rank holding nothing of the chunk
    torch.manual_seed(32)
    hook.NormalPerturb(0.05).apply_batched(src[0:0].clone(), hook.HookSpan(0, 0, 0, n))
    assert torch.equal(after, torch.rand(1))


def test_hooks_of_a_step_follow_the_reference_call_sequence():
    """closure.apply_hooks: one hook call per reference chunk and hooked variable, in the
    reference's order, for any execution batch: ScheduledNormalPerturb's clock and the
    interleaving of two random hooks match the chunk-by-chunk per-sample loop"""
    from pix2latent_amd import VariableManager
    from pix2latent_amd.optimizer.closure import apply_hooks
    from pix2latent_amd.variable_manager import slice_vars
    from pix2latent_amd.utils import function_hooks as hook

    def make():
        vm = VariableManager(device='cpu')
        vm.register('a', (4,), 'input', default=torch.zeros(4),
                    hook_fn=hook.Compose(hook.NormalPerturb(0.1), hook.Clamp(0.15)))
        vm.register('b', (3,), 'input', default=torch.zeros(3),
                    hook_fn=hook.ScheduledNormalPerturb(0.5, max_step=9))
        return vm.initialize(5)
    # reference order: for chunk in chunks of 2: for var in (a, b): hook(chunk's samples)
    ref = make()
    torch.manual_seed(5)
    for step in range(2):
        for lo in (0, 2, 4):
            for name in ('a', 'b'):
                ref.input[name].hook_fn(ref.input[name].data[lo:lo + 2])
    assert ref.input.b.hook_fn.t == 6
    # one pass over the whole population
    one = make()
    torch.manual_seed(5)
    for step in range(2):
        apply_hooks(one, (0, 5), 2)
    assert one.input.b.hook_fn.t == 6
    for name in ('a', 'b'):
        assert torch.equal(one.input[name].buf, torch.stack(list(ref.input[name].data)))
    # a block [1,4) of the population (a rank of a sharded run)
    part = make()
    torch.manual_seed(5)
    for step in range(2):
        apply_hooks(slice_vars(part, 1, 4), (1, 5), 2)
    for name in ('a', 'b'):
        assert torch.equal(part.input[name].buf[1:4], torch.stack(list(ref.input[name].data))[1:4])
        assert torch.count_nonzero(part.input[name].buf[0]) == 0
