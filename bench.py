#!/usr/bin/env python
"""bench.py -- candidate-latent evals/sec of the pix2latent hot path on MI355X.

Workload (BASELINE.json `metric`, configs[2]): BigGAN-deep-256 BasinCMA inner
step, population 18 (pycma popsize for z in R^128), loss = weighted L1 +
10 * weighted LPIPS-VGG16.  One "step" = one inner Adam step over the WHOLE
population: hooks -> generator forward -> loss -> backward to (z, c) -> Adam,
i.e. 18 candidate evals.  On N > 1 GPUs the population is block-sharded over
the ranks (strong scaling: 18 candidates in total, whatever N is).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8 ...

Prints ONE JSON line (rank 0).  Synthetic data: seeded random-init weights of
the exact architectures and a synthetic 256x256 target (no network here).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

POP = 18
MAX_BATCH = 9
FP32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6      # dense v_mfma_f32_32x32x16_bf16 = 16 x the fp32 MFMA rate
# what the matrix pipe SUSTAINS on random fp16 operands, every SIMD multiplying back to back for 10-20 ms: the chip
# clocks to its power budget (tools/micro/mfma_power.hip -> profiles/round6_mfma_power.txt: 2 460 on zeros, 1 670 on
# random operands = 1.59 GHz).  Reported beside `frac` (which stays priced against the nominal dense peak).
MFMA_SUSTAINED_RANDOM_TFLOPS = 1670.0
HBM_PEAK_TBS = 8.0                  # MI355X_MICROARCH.md: HBM3E peak (achievable ~6.3 read; measured write 4.4-4.9)
GFLOP_PER_EVAL = 197.8              # BASELINE.md §2 (conv_to_rgb sliced to 3 channels)


def build_problem(dev, seed=0, exec_batch_size=None, lpips_net='vgg'):
    from pix2latent_amd import VariableManager, distribution
    from pix2latent_amd.utils import synthetic as S, function_hooks as hook
    from pix2latent_amd.model.biggan import BigGAN
    from pix2latent_amd.optimizer import BasinCMAOptimizer
    import pix2latent_amd.loss_functions as LF
    import warnings
    warnings.simplefilter('ignore')
    W = S.biggan_weights(0)
    Wv = {'vgg': lambda: S.lpips_vgg_weights(1), 'alex': lambda: S.lpips_alex_weights(2),
          'squeeze': lambda: S.lpips_squeeze_weights(3)}[lpips_net]()
    model = BigGAN(weights=W, device=dev)
    loss_fn = LF.ProjectionLoss(lpips_net=lpips_net, weights=Wv, device=dev)
    g = torch.Generator().manual_seed(2)
    c_default = 0.05 * torch.randn(128, generator=g)
    target = S.synthetic_target(256, 1)
    weight = S.synthetic_weight_mask(256)
    vm = VariableManager(device=dev)
    # as examples/invert_biggan_basincma.py:61-96 registers them
    vm.register('z', (128,), 'input', distribution=distribution.TruncatedNormalModulo(),
                learning_rate=0.05, hook_fn=hook.Clamp(2.0), grad_free=True)
    vm.register('c', (128,), 'input', default=c_default, learning_rate=0.01)
    vm.register('target', (3, 256, 256), 'output', requires_grad=False, default=target)
    vm.register('weight', (3, 256, 256), 'output', requires_grad=False, default=weight)
    opt = BasinCMAOptimizer(model, vm, loss_fn, max_batch_size=MAX_BATCH,
                            exec_batch_size=exec_batch_size)
    opt.cma_seed = seed
    return opt, vm, (W, Wv, c_default, target, weight)


def _best_threads():
    """torch-CPU convs get slower when all cores of a big host are
    oversubscribed: pick the fastest of a few thread counts on a 1-second probe."""
    import torch.nn.functional as F
    ncpu = os.cpu_count() or 1
    x = torch.randn(2, 64, 128, 128)
    w = torch.randn(64, 64, 3, 3)
    best, best_t = 1, float('inf')
    for n in sorted(set([min(ncpu, k) for k in (8, 16, 32, 64, 128, 256)])):
        torch.set_num_threads(n)
        F.conv2d(x, w, padding=1)
        t0 = time.perf_counter()
        for _ in range(3):
            F.conv2d(x, w, padding=1)
        t = time.perf_counter() - t0
        if t < best_t:
            best, best_t = n, t
    return best


def cpu_baseline(problem, chunk=MAX_BATCH):
    """The oracle (CPU restatement, torch-CPU fp32, best thread count of the host) timed on a
    bounded sample of the same workload: ONE reference chunk of `chunk` (= max_batch_size = 9)
    candidates, fwd + loss + bwd, once with weight gradients enabled -- what the reference
    does: it never freezes the generator / LPIPS parameters, so loss.backward() also computes
    the generator's weight gradients (closure.py:58) -- and once with input gradients only
    (what the native path computes).  `value` is the reference behaviour (wgrad on)."""
    from oracle import biggan_ref as R, lpips_ref as L
    W, Wv, c_default, target, weight = problem
    g = torch.Generator().manual_seed(2)
    torch.set_num_threads(_best_threads())

    def one(n, wgrad):
        # nn.Parameters of the reference's modules: every generator weight / bias / BN affine /
        # gamma (running statistics are buffers); of lpips.LPIPS only the `lin` layers -- its
        # backbone is built with requires_grad=False (pnet_tune=False) [3P-recall]
        Wg = {k: (v.detach().clone().requires_grad_(True)
                  if wgrad and torch.is_floating_point(v) and 'running_' not in k else v)
              for k, v in W.items()}
        Wvg = {k: (v.detach().clone().requires_grad_(True)
                   if wgrad and k.startswith('lpips.lin') else v) for k, v in Wv.items()}
        z = torch.fmod(torch.randn(n, 128, generator=g), 2.0).requires_grad_(True)
        c = c_default.unsqueeze(0).repeat(n, 1).requires_grad_(True)
        t0 = time.perf_counter()
        out = R.biggan_forward(Wg, z, c)
        loss = L.projection_loss(Wvg, out, target.unsqueeze(0).repeat(n, 1, 1, 1),
                                 weight.unsqueeze(0).repeat(n, 1, 1, 1))
        loss.mean().backward()
        return time.perf_counter() - t0
    one(chunk, False)                       # warm-up at the timed shapes (thread pools, oneDNN primitives)
    t_off = one(chunk, False)
    t_on = sorted(one(chunk, True) for _ in range(3))        # median of 3 (one sample moved 0.58-0.75)
    cpu_model = None
    try:
        with open('/proc/cpuinfo') as f:
            cpu_model = next((l.split(':', 1)[1].strip() for l in f if l.startswith('model name')), None)
    except OSError:
        pass
    # BASELINE configs[0] ("C1 ... timed for 5 steps"): examples/invert_biggan_adam.py with
    # num_samples = 1 and the weighted L1 term alone -- fwd + loss + bwd (weight gradients on, as the reference
    # computes them) + torch.optim.Adam on (z, c), lr 0.05 / 0.01 (variable_manager.py:231-235)
    def config1(steps=5):
        Wg = {k: (v.detach().clone().requires_grad_(True)
                  if torch.is_floating_point(v) and 'running_' not in k else v) for k, v in W.items()}
        z = torch.fmod(torch.randn(1, 128, generator=g), 2.0).requires_grad_(True)
        c = c_default.unsqueeze(0).clone().requires_grad_(True)
        adam = torch.optim.Adam([{'params': [z], 'lr': 0.05}, {'params': [c], 'lr': 0.01}])
        times = []
        for _ in range(steps + 1):
            t0 = time.perf_counter()
            adam.zero_grad()
            with torch.no_grad():
                z.clamp_(-2.0, 2.0)
            out = R.biggan_forward(Wg, z, c)
            loss = L.reconstruction_loss(out, target.unsqueeze(0), weight.unsqueeze(0))
            loss.mean().backward()
            adam.step()
            times.append(time.perf_counter() - t0)
        return times[1:]                                     # (the first step warms the primitives up)
    t_c1 = config1()
    return {'value': round(chunk / t_on[1], 4), 'unit': 'evals/s',
            'config1': {'evals_per_s': round(len(t_c1) / sum(t_c1), 4), 'ms_per_step': round(1e3 * sum(t_c1) / len(t_c1), 1),
                        'sample': '5 steps of BASELINE config 1 (num_samples = 1, weighted L1 only, BigGAN-deep-256 fwd + '
                                  'bwd with weight gradients + Adam), after one warm-up step',
                        'gpu_side': 'config.extra.biggan_gradient_n1_l1'},
            'cores': torch.get_num_threads(), 'kind': 'port',
            'cpu': cpu_model, 'host_cpus': os.cpu_count(),
            'samples_evals_per_s': [round(chunk / t, 4) for t in t_on],
            'wgrad': {'on': round(chunk / t_on[1], 4), 'off': round(chunk / t_off, 4)},
            'sample': '1 reference chunk of %d candidates (fwd+loss+bwd, BigGAN-deep-256 + '
                      'L1+10*LPIPS-VGG16, torch-CPU fp32), weight gradients enabled (the reference '
                      'never freezes the networks): median of 3 timings = value; once with input '
                      'gradients only' % chunk}


class _GpuTelemetry(object):
    """socket power and shader clock of the bench GPU while the timed steps run, read from the
    amdgpu hwmon files at ~4 Hz on a side thread (20 Hz cost the timed steps 0.4 %: every read is an SMU query) (no rocm-smi process; null when the files are
    not there).  The kernels of this step are power-limited: the clock a number was
    measured at belongs next to the number."""

    def __init__(self, index=0, enabled=True):
        import glob
        self.paths = None
        cards = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')) if enabled else []
        cards = [c for c in cards if os.path.exists(os.path.join(c, 'freq1_input'))]
        # the box shows every GPU of the node in sysfs; the one this process runs on is found by
        # its PCI address
        try:
            pr = torch.cuda.get_device_properties(index)
            want = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            mine = [c for c in cards
                    if os.path.basename(os.path.realpath(os.path.join(c, '..', '..'))) == want]
            cards = mine or cards
        except Exception:           # noqa: BLE001  (older torch: no PCI fields)
            pass
        if cards:
            h = cards[0 if len(cards) == 1 else min(index, len(cards) - 1)]
            pw = next((os.path.join(h, n) for n in ('power1_average', 'power1_input')
                       if os.path.exists(os.path.join(h, n))), None)
            self.paths = (pw, os.path.join(h, 'freq1_input'))
        self.power, self.sclk, self._stop = [], [], False

    def _read(self, path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError, TypeError):
            return None

    def _loop(self):
        while not self._stop:
            pw, fq = self._read(self.paths[0]), self._read(self.paths[1])
            if pw is not None:
                self.power.append(pw * 1e-6)
            if fq is not None:
                self.sclk.append(fq * 1e-6)
            time.sleep(0.25)

    def __enter__(self):
        if self.paths is not None:
            import threading
            self._t = threading.Thread(target=self._loop, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self.paths is not None:
            self._t.join(1.0)

    def summary(self):
        def stat(v):
            return None if not v else {'min': round(min(v), 1), 'mean': round(sum(v) / len(v), 1),
                                       'max': round(max(v), 1), 'samples': len(v)}
        return {'socket_power_w': stat(self.power), 'sclk_mhz': stat(self.sclk),
                'source': 'amdgpu hwmon (power1_average, freq1_input) sampled during the timed steps'
                          if self.paths else None}


# ---------------------------------------------------------------------------------------
# the other configurations BASELINE.json / north_star name, measured AFTER the timed region
# and reported under config.extra (same JSON line)
# ---------------------------------------------------------------------------------------
def _time_steps(fn, n_warm, n, reps=3):
    """seconds per step: the MEDIAN of `reps` timings of n steps each (three steps alone caught a
    30 ms host stall once in a dozen runs: 430 instead of 940 evals/s for the 8-sample configuration)"""
    for i in range(n_warm):
        fn(i == 0)
    per = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn(False)
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / n)
    return sorted(per)[len(per) // 2]


def extra_configs(dev, n_steps=6):
    import contextlib
    import warnings
    warnings.simplefilter('ignore')
    from pix2latent_amd import VariableManager, distribution
    from pix2latent_amd.utils import synthetic as S, function_hooks as hook
    from pix2latent_amd.model.biggan import BigGAN
    from pix2latent_amd.model.stylegan2 import StyleGAN2
    from pix2latent_amd.optimizer import GradientOptimizer
    import pix2latent_amd.loss_functions as LF
    out = {}
    Wv = S.lpips_vgg_weights(1)

    def run(name, model, vm, n, size, note, profile=None, loss_fn=None, **kw):
        if loss_fn is None:
            loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=Wv, device=dev)
        opt = GradientOptimizer(model, vm, loss_fn, max_batch_size=MAX_BATCH, **kw)
        variables = vm.initialize(num_samples=n)
        # (4 untimed steps: with <= 6 candidates the step is captured in a HIP graph on its third call)
        dt = _time_steps(lambda first: opt.step(variables, optimize=True, transform=first), 4, n_steps)
        losses = [float(x) for x in opt.loss]
        assert all(l == l for l in losses), name
        out[name] = {'evals_per_s': round(n / dt, 1), 'ms_per_step': round(1e3 * dt, 2),
                     'candidates': n, 'resolution': size, 'what': note,
                     'hip_graph_replay': bool(opt._graphs) and any(isinstance(v, tuple) for v in opt._graphs.values())}
        if profile:
            # which kernel dominates and where it stands: the 3x3 conv launches of two more
            # (eager) steps timed with the library's hipEvent pairs, every 4th launch
            from pix2latent_amd import _native as N
            lib = N.lib()
            saved, opt.use_graph = opt.use_graph, False
            N.check(lib.p2l_prof_begin(8192), 'p2l_prof_begin')
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(4):
                lib.p2l_prof_step(i, 4)
                opt.step(variables, optimize=True)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            T = N.prof_totals()
            f, m, c, b, x, mf = T.flops, T.ms, T.count, T.bytes, T.exec_flops, T.mfma_flops
            opt.use_graph = saved
            if m[0] > 0:
                tf = f[0] / (m[0] * 1e-3) / 1e12
                dom = max(FAM_3X3, key=lambda q: T.fam_ms[q])
                out[name]['dominant_kernel'] = {
                    'family': '3x3 / sub-pixel convs (wino16s_conv_kernel + conv_h2_kernel<9|4>)',
                    'by_time': {N.PROF_FAMILIES[q]: round(T.fam_ms[q], 3) for q in range(7) if T.fam_count[q]},
                    'kernel': N.PROF_FAMILIES[dom],
                    'kernel_avg_launch_ms': round(T.fam_ms[dom] / max(T.fam_count[dom], 1), 4),
                    'kernel_frac_executed_of_bf16_peak': round(
                        T.fam_mfma_flops[dom] / (T.fam_ms[dom] * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 3)
                    if T.fam_ms[dom] > 0 else None,
                    'time_share_of_step': round(4 * m[0] * 1e-3 / el, 3),
                    'achieved_tflops_algorithmic': round(tf, 1),
                    'frac_of_bf16x3_ceiling': round(tf / (BF16_MFMA_PEAK_TFLOPS / 6), 3),
                    'mfma_products_per_fp32_product': round(mf[0] / x[0], 2) if x[0] > 0 else None,
                    'frac_executed_of_bf16_peak': round(mf[0] / (m[0] * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 3),
                    'rocprof': profile}
        del opt, variables, loss_fn
        torch.cuda.empty_cache()

    with contextlib.redirect_stdout(sys.stderr):
        # C2: BigGAN-256 GradientOptimizer, num_samples = 8 (one chunk)
        vm = VariableManager(device=dev)
        g = torch.Generator().manual_seed(2)
        vm.register('z', (128,), 'input', distribution=distribution.TruncatedNormalModulo(),
                    learning_rate=0.05, hook_fn=hook.Clamp(2.0))
        vm.register('c', (128,), 'input', default=0.05 * torch.randn(128, generator=g),
                    learning_rate=0.01)
        vm.register('target', (3, 256, 256), 'output', requires_grad=False, default=S.synthetic_target(256, 1))
        vm.register('weight', (3, 256, 256), 'output', requires_grad=False, default=S.synthetic_weight_mask(256))
        biggan = BigGAN(weights=S.biggan_weights(0), device=dev)
        run('biggan_gradient_n8', biggan, vm, 8, 256,
            'BASELINE config 2: GradientOptimizer inner step, 8 samples, L1 + 10*LPIPS-VGG16')
        # C1: examples/invert_biggan_adam.py with num_samples = 1 and the L1 term alone (BASELINE configs[0], the
        # reference's own CPU-runnable case; its CPU timing is cpu_baseline.config1)
        run('biggan_gradient_n1_l1', biggan, vm, 1, 256,
            'BASELINE config 1: examples/invert_biggan_adam.py:19-31,108 with num_samples = 1, weighted L1 only '
            '(ReconstructionLoss): one candidate per step, launch-latency bound',
            loss_fn=LF.ReconstructionLoss())
        del biggan

        # C4: StyleGAN2 cars 512^2, 32 samples, Compose(NormalPerturb, Clamp), loss mask
        gen = StyleGAN2(model='cars', search='z', device=dev)
        fixed = [torch.randn(1, 1, s_[2], s_[3], generator=g).to(dev) for s_ in gen.noise_shape]

        class FixedNoise(torch.nn.Module):
            def forward(self, z=None):
                return gen.forward_z(z, noises=[n_.expand(z.size(0), -1, -1, -1).contiguous()
                                                for n_ in fixed])
        vm = VariableManager(device=dev)
        vm.register('z', (512,), 'input', distribution=distribution.TruncatedNormalModulo(),
                    learning_rate=0.05,
                    hook_fn=hook.Compose(hook.NormalPerturb(sigma=0.05), hook.Clamp(2.0)))
        mask = torch.zeros(3, 512, 512)
        mask[:, 64:-64, :] += 1.0
        for nm, t in (('target', S.synthetic_target(512, 1)), ('weight', torch.ones(3, 512, 512)),
                      ('loss_mask', mask)):
            vm.register(nm, (3, 512, 512), 'output', requires_grad=False, default=t)
        run('stylegan2_cars_512_n32', FixedNoise(), vm, 32, 512,
            'BASELINE config 4 inner step: 32 samples (reference chunks 9,9,9,5 define the '
            'gradient scale; executed in one device pass), z-space, rows 64:-64 loss mask',
            profile='profiles/round6_sg2_512_kernel_stats.csv, round6_sg2_512_layers.txt',
            exec_batch_size='all')
        del gen, fixed
        torch.cuda.empty_cache()

        # C5 shard: StyleGAN2 FFHQ 1024^2 in W+, 3 candidates (pop 22 over 8 ranks), w+ and noises
        gen = StyleGAN2(model='ffhq', search='w+', device=dev)
        n_noise = sum(s_[-2] * s_[-1] for s_ in gen.noise_shape)
        vm = VariableManager(device=dev)
        vm.register('z', (18, 512), 'input', learning_rate=0.05,
                    default=gen.latent_mean.cpu().view(1, 512).repeat(18, 1))
        vm.register('noises', (n_noise,), 'input', learning_rate=0.05,
                    default=torch.randn(n_noise, generator=g))
        vm.register('target', (3, 1024, 1024), 'output', requires_grad=False, default=S.synthetic_target(1024, 1))
        vm.register('weight', (3, 1024, 1024), 'output', requires_grad=False, default=S.synthetic_weight_mask(1024))
        run('stylegan2_ffhq_1024_shard3_wplus', gen, vm, 3, 1024,
            'BASELINE config 5, one rank\'s shard: 3 candidates, W+ latents [18,512] and the '
            '2.8M-element noise vector both optimised',
            profile='profiles/round6_sg2_1024_kernel_stats.csv, round6_sg2_1024_layers.txt')
    return out


def full_schedule_and_default_loss(dev, ms_per_step):
    """Two side measurements of the headline problem, outside the timed region:

    * `full_basincma_30x30_300`: ONE complete run of the schedule the reference's example hard-codes
      (examples/invert_biggan_basincma.py:108-109: optimize(meta_steps=30, grad_steps=30,
      last_grad_steps=300) = 1 200 inner steps = 21 600 candidate evals + 30 re-scores of 18), with
      track() on as the reference has it (base_optimizer.py:100-107): wall seconds, evals/s over the
      whole run, what a generation costs on top of its inner steps (ask, fresh variables + Adam state,
      re-score, tell, tracking copies), and the mean loss the first `tell` saw next to the final one.
    * `biggan_basincma_alex`: the same inner step with ProjectionLoss() as every unmodified example
      constructs it (loss_functions.py:89: lpips_net='alex')."""
    import contextlib
    out = {}
    with contextlib.redirect_stdout(sys.stderr):
        torch.manual_seed(0)
        opt, vm, _ = build_problem(dev, exec_batch_size=MAX_BATCH)
        opt.show_iter = 10 ** 9
        told = []
        score = opt.losses_for_tell

        def recording(variables):
            t = score(variables)
            told.append(float(np.mean(t)))
            return t
        opt.losses_for_tell = recording
        opt.optimize(meta_steps=1, grad_steps=2, last_grad_steps=2)      # (warm-up: workspaces, caches)
        del told[:]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, losses = opt.optimize(meta_steps=30, grad_steps=30, last_grad_steps=300)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        final = [float(x) for x in losses[-1][1]['loss']]
        n_steps = 30 * 30 + 300
        # the run's own steady-state step: 300 inner steps of ONE generation (same tracking, same loop)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        opt.optimize(meta_steps=0, grad_steps=0, last_grad_steps=300)
        torch.cuda.synchronize()
        steady_ms = (time.perf_counter() - t1) / 300 * 1e3
        assert len(told) == 30, len(told)
        assert np.mean(final) < told[0], 'the run did not reduce the loss (%g -> %g)' % (told[0], np.mean(final))
        out['full_basincma_30x30_300'] = {
            'what': 'opt.optimize(meta_steps=30, grad_steps=30, last_grad_steps=300) on the bench problem, '
                    'track() on (reference examples/invert_biggan_basincma.py:102-109)',
            'wall_s': round(wall, 3), 'inner_steps': n_steps, 'candidate_evals': POP * n_steps,
            'rescored_candidates': POP * 30,
            'evals_per_s_over_the_run': round(POP * n_steps / wall, 1),
            'ms_per_step_of_the_timed_region': round(ms_per_step, 3),
            'ms_per_step_inside_a_generation': round(steady_ms, 3),
            'per_generation_overhead_ms': round((wall - n_steps * steady_ms * 1e-3) / 30 * 1e3, 2),
            'overhead_share_of_the_run': round((wall - n_steps * steady_ms * 1e-3) / wall, 4),
            'overhead_is': 'wall - 1200 x (one generation of 300 steps / 300), per CMA generation: ask, fresh '
                           'variables + Adam state, re-score, tell',
            'tracking_and_loop_cost_per_step_ms': round(steady_ms - ms_per_step, 3),
            'mean_loss_first_tell': round(told[0], 5), 'mean_loss_last_tell': round(told[-1], 5),
            'mean_loss_final': round(float(np.mean(final)), 5), 'best_loss_final': round(min(final), 5),
            'tracked_steps': len(opt.tracked['z'])}
        del opt, vm
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        opt, vm, _ = build_problem(dev, exec_batch_size=MAX_BATCH, lpips_net='alex')
        opt.setup_cma(vm)
        variables = opt.cma_init(vm)
        dt = _time_steps(lambda first: opt.step(variables, optimize=True, transform=first), 3, 5)
        losses = [float(x) for x in opt.loss]
        assert all(l == l for l in losses)
        out['biggan_basincma_alex'] = {
            'what': 'the headline inner step with the examples\' default loss, ProjectionLoss(lpips_net=\'alex\') '
                    '(reference loss_functions.py:89): L1 + 10*LPIPS-AlexNet',
            'evals_per_s': round(POP / dt, 1), 'ms_per_step': round(1e3 * dt, 2), 'candidates': POP}
        del opt, vm
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        opt, vm, _ = build_problem(dev, exec_batch_size=MAX_BATCH, lpips_net='squeeze')
        opt.setup_cma(vm)
        variables = opt.cma_init(vm)
        dt = _time_steps(lambda first: opt.step(variables, optimize=True, transform=first), 3, 5)
        assert all(float(l) == float(l) for l in opt.loss)
        out['biggan_basincma_squeeze'] = {
            'what': 'the headline inner step with ProjectionLoss(lpips_net=\'squeeze\'), the third network '
                    'lpips.LPIPS(net=...) takes (reference loss_functions.py:131): L1 + 10*LPIPS-SqueezeNet1.1',
            'evals_per_s': round(POP / dt, 1), 'ms_per_step': round(1e3 * dt, 2), 'candidates': POP}
    return out


def alone_leg(opt, variables, lib, steps=8, period=4):
    """the conv launches of the step with nothing beside them: one pass of the whole population on one stream
    (exec_batch_size = population), `steps` more steps right after the timed region, every launch timed
    steps / period times (hipEvent pairs on the launch stream, phase rotating with the step)"""
    from pix2latent_amd import _native as N
    saved = opt.exec_batch_size
    opt.exec_batch_size = POP                            # (an execution pass above the reference chunk: one stream)
    try:
        opt.step(variables, optimize=True)               # (workspaces of the 18-candidate pass)
        torch.cuda.synchronize()
        N.check(lib.p2l_prof_begin(4096), 'p2l_prof_begin')
        t0 = time.perf_counter()
        for i in range(steps):
            lib.p2l_prof_step(i, period)
            opt.step(variables, optimize=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        T = N.prof_totals()
    finally:
        opt.exec_batch_size = saved
    return T, el, steps, period


FAM_3X3 = (1, 2, 6, 3, 4, 0)          # P2L_PROF_FAM_*: the families a 3x3 launch can belong to


def family_record(T, steps, elapsed, period, bf3, exec_batch, lanes):
    """the roofline record of one measured leg from the library's per-launch totals (P2LProfTotals): the 3x3
    family as a whole (achieved / peak / frac = what the matrix pipe EXECUTES), its DOMINANT KERNEL by time
    (the row to hold against a rocprofv3 --kernel-trace --stats table), the 1x1 family against the memory roof"""
    from pix2latent_amd import _native as N
    flops, ms, cnt, abytes, xflops, mflops, wbytes = (T.flops, T.ms, T.count, T.bytes, T.exec_flops,
                                                      T.mfma_flops, T.write_bytes)
    peak = BF16_MFMA_PEAK_TFLOPS if bf3 else FP32_MFMA_PEAK_TFLOPS
    sec = ms[0] * 1e-3
    conv_tflops = flops[0] / sec / 1e12 if sec > 0 else 0.0
    exec_tflops = xflops[0] / sec / 1e12 if sec > 0 else 0.0
    mfma_tflops = mflops[0] / sec / 1e12 if sec > 0 else 0.0
    prod_mix = mflops[0] / xflops[0] if xflops[0] > 0 else 6.0
    conv1_tflops = flops[1] / (ms[1] * 1e-3) / 1e12 if ms[1] > 0 else 0.0
    ach = mfma_tflops if bf3 else exec_tflops
    dom = max(FAM_3X3, key=lambda f: T.fam_ms[f])
    dsec = T.fam_ms[dom] * 1e-3
    dn = max(T.fam_count[dom], 1)
    dom_ach = T.fam_mfma_flops[dom] / dsec / 1e12 if dsec > 0 else 0.0
    rec = {
        'exec_batch_size': exec_batch, 'lanes': lanes,
        'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
        'frac_is': 'executed 16-bit MFMA FLOP/s / dense 16-bit MFMA peak' if bf3
                   else 'executed fp32 MFMA FLOP/s / fp32 MFMA peak',
        'frac_of_sustained': ({'value': round(ach / MFMA_SUSTAINED_RANDOM_TFLOPS, 4),
                               'sustained_tflops_random_fp16_operands': MFMA_SUSTAINED_RANDOM_TFLOPS,
                               'source': 'profiles/round6_mfma_power.txt (tools/micro/mfma_power.hip): back-to-back '
                                         'v_mfma_f32_32x32x16_f16 on every SIMD, power-limited clock 1.59 GHz; the '
                                         'nominal peak is reached on zero operands only'} if bf3 else None),
        'avg_launch_ms': round(ms[0] / max(cnt[0], 1), 4),
        'sampled_launches': int(cnt[0]),
        'launches_per_step': round(cnt[0] * period / steps, 1),
        'time_share_of_step': round(period * ms[0] * 1e-3 / elapsed, 4),
        'ms_per_step_of_this_leg': round(1e3 * elapsed / steps, 3),
        'algo_bytes_per_launch': round(abytes[0] / max(cnt[0], 1)),
        'dominant_kernel': {
            'name': N.PROF_FAMILIES[dom],
            'launches_per_step': round(T.fam_count[dom] * period / steps, 1),
            'avg_launch_ms': round(T.fam_ms[dom] / dn, 4),
            'time_share_of_step': round(period * dsec / elapsed, 4),
            'achieved': round(dom_ach, 1), 'frac': round(dom_ach / peak, 4),
            'algorithmic_tflops': round(T.fam_flops[dom] / dsec / 1e12, 1) if dsec > 0 else None,
            'gflop_per_launch': round(T.fam_flops[dom] / dn / 1e9, 2),
            'algo_bytes_per_launch': round(T.fam_bytes[dom] / dn),
            'check': 'avg_launch_ms x launches = the Calls x AverageNs row of this kernel in the rocprofv3 '
                     '--kernel-trace --stats table of the same configuration (profiles/round6_kernel_stats_*.csv)'},
        'families_ms_per_step': {N.PROF_FAMILIES[f]: round(T.fam_ms[f] * period / steps, 3)
                                 for f in range(7) if T.fam_count[f]},
        # the algorithmic (fp32-equivalent, 9 taps on the output grid, 3 real image channels) rate of the
        # same launches and its ratios
        'algorithmic': {
            'tflops': round(conv_tflops, 2),
            'gflop_per_launch': round(flops[0] / max(cnt[0], 1) / 1e9, 3),
            'vs_fp32_mfma_peak': round(conv_tflops / FP32_MFMA_PEAK_TFLOPS, 4),
            'mfma_products_per_fp32_product': round(prod_mix, 3),
            'vs_mix_ceiling': round(conv_tflops / (BF16_MFMA_PEAK_TFLOPS / prod_mix), 4) if bf3 else None,
            'ceiling_of_that_ratio': round(flops[0] / xflops[0], 3) if xflops[0] > 0 else None,
            'vs_bf16x3_ceiling': round(conv_tflops / (BF16_MFMA_PEAK_TFLOPS / 6), 4) if bf3 else None,
            'executed_fp32_equiv_tflops': round(exec_tflops, 2)},
        # the 1x1 family is output-dominated: its roof is memory, and the WRITE rate of the part (4.4-4.9
        # TB/s, tools/micro/mem_rate.hip) rather than the 8 TB/s headline
        'conv1x1': {'achieved': round(conv1_tflops, 2), 'unit': 'TFLOP/s',
                    'mfma_products_per_fp32_product': round(mflops[1] / xflops[1], 2) if xflops[1] > 0 else None,
                    'sampled_launches': int(cnt[1]),
                    'launches_per_step': round(cnt[1] * period / steps, 1),
                    'time_share_of_step': round(period * ms[1] * 1e-3 / elapsed, 4),
                    'bound': 'hbm',
                    'achieved_tb_per_s': round(abytes[1] / (ms[1] * 1e-3) / 1e12, 3) if ms[1] > 0 else None,
                    'peak_tb_per_s': HBM_PEAK_TBS,
                    'measured_stream_tb_per_s': {'write': 4.5, 'read': 6.5, 'copy': 5.0},
                    'frac': round(abytes[1] / (ms[1] * 1e-3) / 1e12 / HBM_PEAK_TBS, 4) if ms[1] > 0 else None,
                    'per_layer_table': 'profiles/round6_conv1x1_roofline.txt'},
    }
    # PMC counters cannot be read from inside the timed process: `traffic` is the committed result of the
    # separate rocprofv3 --pmc passes over `bench.py --pmc-run --exec-batch <this leg's>` (tools/gpu_round6.sh
    # -> tools/traffic_json.py), taken at THIS leg's execution batch -- a file from another batch is refused
    traffic, src, rw = None, None, None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'round6_traffic.json')
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        run = next((r for r in tj.get('runs', []) if r.get('exec_batch_size') == exec_batch), None)
        if run is None:
            src = {'file': 'profiles/round6_traffic.json',
                   'refused': 'no PMC pass at exec_batch_size %s (have %s)'
                              % (exec_batch, [r.get('exec_batch_size') for r in tj.get('runs', [])])}
        elif cnt[0] > 0:
            a_wr = wbytes[0] / cnt[0]
            a_rd = abytes[0] / cnt[0] - a_wr
            traffic = run['hbm_bytes_per_launch']
            rw = {'fetch_bytes_per_launch': run['fetch_bytes_per_launch'],
                  'write_bytes_per_launch': run['write_bytes_per_launch'],
                  'algo_read_bytes_per_launch': round(a_rd), 'algo_write_bytes_per_launch': round(a_wr),
                  'fetch_over_algorithmic': round(run['fetch_bytes_per_launch'] / a_rd, 3) if a_rd > 0 else None,
                  'write_over_algorithmic': round(run['write_bytes_per_launch'] / a_wr, 3) if a_wr > 0 else None,
                  'traffic_over_algorithmic': round(traffic / (a_rd + a_wr), 3) if a_rd + a_wr > 0 else None}
            src = {'file': 'profiles/round6_traffic.json', 'commit': tj.get('commit'), 'box': tj.get('box'),
                   'command': run.get('command'), 'exec_batch_size': run.get('exec_batch_size'),
                   'launches_counted': run.get('fetch_launches'),
                   'note': 'separate rocprofv3 --pmc passes (FETCH_SIZE x 2 per the guide, WRITE_SIZE) over the '
                           'bench command at this execution batch'}
    rec['traffic'] = traffic
    rec['traffic_unit'] = 'HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)'
    rec['traffic_source'] = src
    rec['traffic_read_write'] = rw
    return rec


def exact_fp32_leg(dev, args):
    """evals/s of the same inner step with P2L_CONV_WFMT=f32 (outside the timed K steps)."""
    import contextlib
    prev = os.environ.get('P2L_CONV_WFMT')
    os.environ['P2L_CONV_WFMT'] = 'f32'
    try:
        torch.manual_seed(0)
        opt, vm, _ = build_problem(dev, exec_batch_size=args.exec_batch, lpips_net=args.lpips_net)
        with contextlib.redirect_stdout(sys.stderr):
            opt.setup_cma(vm)
            variables = opt.cma_init(vm)
        for i in range(2):
            opt.step(variables, optimize=True, transform=(i == 0))
        torch.cuda.synchronize()
        n = max(2, min(args.steps, 5))
        t0 = time.perf_counter()
        for _ in range(n):
            opt.step(variables, optimize=True)
        torch.cuda.synchronize()
        return POP * n / (time.perf_counter() - t0)
    finally:
        if prev is None:
            os.environ.pop('P2L_CONV_WFMT', None)
        else:
            os.environ['P2L_CONV_WFMT'] = prev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--prof-period', type=int, default=20,
                    help='time every N-th conv launch with hipEvents inside the timed steps '
                         '(1 = every launch; costs ~5 %% of the step)')
    ap.add_argument('--no-telemetry', action='store_true',
                    help='do not sample socket power / shader clock while the timed steps run')
    ap.add_argument('--no-fp32-leg', action='store_true',
                    help='skip the extra exact-fp32-MFMA measurement reported in config')
    ap.add_argument('--eager', action='store_true',
                    help='run the timed steps eagerly with the per-launch hipEvent pairs inside them (rounds 1-5). '
                         'Default: the timed steps run as the optimizers run them -- for this configuration one HIP '
                         'graph with two branches, replayed -- and the launches are timed in an eager leg behind them')
    ap.add_argument('--no-alone', action='store_true',
                    help='skip the one-pass-of-18 leg that measures the kernels without the other lane beside them')
    ap.add_argument('--pmc-run', action='store_true',
                    help='profiling runs (rocprofv3 --pmc / --kernel-trace): warm-up + timed steps ONLY -- no re-score, '
                         'no alone leg, no side configurations -- so that every dispatch the tool sees belongs to '
                         'the configuration named on the command line')
    ap.add_argument('--no-extra', action='store_true',
                    help='skip the additional configurations reported under config.extra')
    ap.add_argument('--lpips-net', default='vgg', choices=['vgg', 'alex', 'squeeze'],
                    help="LPIPS network: 'vgg' = BASELINE.json's metric (default); 'alex' = the "
                         "reference's ProjectionLoss() default, reported as a side configuration")
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend: 'nccl' (= RCCL) or "
                    "'gloo' (test only: lets several ranks share one GPU)")
    ap.add_argument('--exec-batch', type=int, default=MAX_BATCH,
                    help='candidates pushed through the device per pass (semantic chunk size stays '
                         'max_batch_size=9).  9 (default) = the reference chunks, which the fused step runs '
                         'as two lanes on two HIP streams (pix2latent_amd/lanes.py; P2L_STREAMS=1: one '
                         'after the other); 18 = one pass of the whole population on one stream')
    args = ap.parse_args()
    if args.pmc_run:
        args.eager = args.no_alone = args.no_fp32_leg = args.no_extra = args.no_cpu_baseline = True

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (ROCm device); none visible')
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    from pix2latent_amd import _native as N
    torch.manual_seed(0)
    opt, vm, problem = build_problem(dev, exec_batch_size=args.exec_batch, lpips_net=args.lpips_net)
    # stdout carries exactly ONE JSON line: the optimizers' informational prints
    # ("(cma-es) number of samples: 18", reference base_cma_optimizer.py:56) go to stderr
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        opt.setup_cma(vm)
        assert opt.num_samples == POP
        variables = opt.cma_init(vm)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The timed steps run the step as the optimizers run it (`opt.optimize()`): for 18 candidates in two reference
    # chunks that is ONE HIP graph with two branches, replayed (base_optimizer._graph_default; with <= 6 local
    # candidates, i.e. on the ranks of a 4- / 8-GPU job, a one-branch graph).  A captured graph cannot contain
    # hipEvent pairs, so the per-launch figures come from legs BEHIND the timed region: an eager leg in the same
    # two-lane configuration (`roofline.concurrent`) and the one-pass-of-18 leg (top-level `roofline`).
    # --eager: the rounds 1-5 form, events inside the timed steps.
    graph_timed = not args.eager
    if not graph_timed:
        opt.use_graph = False
    for i in range(args.warmup):
        opt.step(variables, optimize=True, transform=(i == 0))

    def replaying():
        return any(isinstance(v, tuple) for v in getattr(opt, '_graphs', {}).values())
    if graph_timed:
        # (a graph is captured on the second sighting of a set of buffers: with --warmup < 3 that would fall
        #  into the timed steps; this is set-up, like allocating the workspaces)
        extra_warm = 0
        while not replaying() and extra_warm < 3 and opt.use_graph is not False:
            opt.step(variables, optimize=True, transform=(args.warmup == 0 and extra_warm == 0))
            extra_warm += 1
    sync()
    lanes_used = max(1, len(getattr(opt.model, '_lanes', {})))    # (sets of workspaces the model was asked for)
    lib = N.lib()
    n_prof = 4096
    # hipEvents around EVERY conv launch cost the stream ~5 us each (measured: 5 % of the
    # step); time every PERIOD-th launch, rotating the phase with the step, so that each
    # launch of the step is timed once per PERIOD steps
    period = max(1, min(args.prof_period, args.steps))
    if not graph_timed:
        N.check(lib.p2l_prof_begin(n_prof), 'p2l_prof_begin')
    with _GpuTelemetry(local_rank, enabled=not args.no_telemetry) as telemetry:
        t0 = time.perf_counter()
        for i in range(args.steps):
            if not graph_timed:
                lib.p2l_prof_step(i, period)
            opt.step(variables, optimize=True)
        sync()
        elapsed = time.perf_counter() - t0
    graph_replayed = graph_timed and replaying()
    if graph_timed:
        # the launches of the same two-lane step, eagerly, with event pairs: the leg `roofline.concurrent` (or,
        # without lanes, the top-level record) is built from
        saved_graph, opt.use_graph = opt.use_graph, False
        ev_steps = max(period, min(args.steps, 20))
        opt.step(variables, optimize=True)
        sync()
        N.check(lib.p2l_prof_begin(n_prof), 'p2l_prof_begin')
        t0e = time.perf_counter()
        for i in range(ev_steps):
            lib.p2l_prof_step(i, period)
            opt.step(variables, optimize=True)
        sync()
        ev_elapsed = time.perf_counter() - t0e
        opt.use_graph = saved_graph
    else:
        ev_steps, ev_elapsed = args.steps, elapsed
    T = N.prof_totals()
    last_loss = [float(x) for x in opt.loss]     # (sharded: the one all-gather, on every rank)
    # SURVEY 8(d) also asks for the fwd-only rate (the CMA re-score); outside the timed K steps
    sync()
    t1 = time.perf_counter()
    n_rescore = 0 if args.pmc_run else 3
    for _ in range(n_rescore):
        opt.step(variables, optimize=False)
    sync()
    rescore_rate = POP * n_rescore / (time.perf_counter() - t1) if n_rescore else 0.0

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        evals = POP * args.steps
        # generator fwd+dgrad 58.80 GMAC + LPIPS net fwd+dgrad (VGG16 40.08 | AlexNet 1.74 GMAC)
        gflop_eval = GFLOP_PER_EVAL if args.lpips_net == 'vgg' else 2 * (58.80 + 1.737)
        bf3 = N.default_wfmt() != N.WFMT_F32
        # what the timed steps measured.  With two lanes a launch's hipEvent pair also spans the time it
        # shares the GPU with the OTHER lane's launches: the step is faster, every launch looks slower, and
        # the durations sum to more than the step -- reported, but not as the kernel's figure
        timed = family_record(T, ev_steps, ev_elapsed, period, bf3, args.exec_batch, lanes_used)
        if world > 1:
            # a rank's launches hold its block of the population (2 ... 9 candidates), not an execution batch the
            # single-GPU PMC passes were taken at: no traffic figure rather than one that belongs to other launches
            timed['traffic'] = timed['traffic_read_write'] = None
            timed['traffic_source'] = {'refused': 'PMC passes exist for single-GPU execution batches only; this '
                                                  'rank launches its block of %d ranks' % world}
            timed['exec_batch_size'] = 'rank-local block (population %d over %d ranks)' % (POP, world)
        kernel_text = (
            'every 3x3 conv launch of the step: wino16s_conv_kernel<.., H2> (Winograd F(2x2,3x3), '
            '16x16-pixel blocks, hand-scheduled; fp16 x 2 arithmetic: power-of-two scaled operands '
            'in two fp16 pieces, 3 x v_mfma_f32_32x32x16_f16 per product; per-image maxima handed '
            'over by the launch that wrote the input, wino_amax_kernel where none did), '
            'conv_h2_kernel<TAPS=9|4> (direct | sub-pixel, the same fp16 x 2 arithmetic) and '
            'conv_thinin/thinout_kernel (3-channel image convs) in the bf16 x 3 arithmetic (6 x '
            'v_mfma_f32_32x32x16_bf16 per product on 3-way split fp32 operands); `dominant_kernel` names the one '
            'that takes the most time' if bf3 else
            'conv_mfma_kernel<TAPS=9> (3x3 implicit GEMM, v_mfma_f32_32x32x2_f32)')
        # achieved / peak / frac are what the matrix pipe EXECUTES -- 16-bit MFMA
        # FLOP/s issued by these launches (3 products per fp32 product for the fp16 x 2 launches, 6 for bf16
        # x 3; Winograd launches 16 instead of 36 products per output quad, sub-pixel launches 4
        # phase-taps, image convs their padded channels) over the dense 16-bit MFMA peak.
        roof = {'kernel': kernel_text, 'bound': 'mfma'}
        if lanes_used > 1 and world == 1 and not args.no_alone:
            # the TOP-LEVEL figures are the kernels' own -- the same build, the same
            # launches with nothing beside them: one pass of the 18 candidates on one stream, 8 more steps
            # right after the timed region, every launch timed twice.  What the launches get while the two
            # lanes overlap is under `concurrent`.
            Ta, el_a, st_a, per_a = alone_leg(opt, variables, lib)
            roof.update(family_record(Ta, st_a, el_a, per_a, bf3, POP, 1))
            roof['measured'] = ('one pass of %d candidates on one stream (exec_batch_size = population), %d steps '
                                'right after the timed region, hipEvent pairs on the launch stream around every '
                                '%d-th conv launch, phase rotating with the step: NON-overlapped launch durations. '
                                'The timed steps themselves run the reference chunks on %d lanes: see `concurrent`'
                                % (POP, st_a, per_a, lanes_used))
            conc = {k: timed[k] for k in ('exec_batch_size', 'lanes', 'achieved', 'frac', 'avg_launch_ms',
                                          'sampled_launches', 'launches_per_step', 'time_share_of_step',
                                          'algo_bytes_per_launch', 'traffic', 'traffic_source', 'traffic_read_write')}
            conc['dominant_kernel'] = timed['dominant_kernel']
            conc['conv1x1'] = timed['conv1x1']
            conc['what'] = ('the launches of the timed configuration (reference chunks of %d on %d HIP streams), %s: a '
                            'hipEvent pair spans the time a launch shares the GPU with the other lane, so these '
                            'durations overlap -- time_share_of_step sums overlapping intervals and may exceed 1 -- '
                            'and `achieved` is what a launch gets while overlapped, not what the kernel does'
                            % (args.exec_batch, lanes_used,
                               ('%d eager steps right behind the timed region (%.3f ms per step; the timed steps '
                                'replay a HIP graph, which cannot hold event pairs)'
                                % (ev_steps, 1e3 * ev_elapsed / ev_steps)) if graph_timed else 'inside the timed steps'))
            roof['concurrent'] = conc
        else:
            roof.update(timed)
            roof['measured'] = (('%d eager steps right behind the timed region' % ev_steps if graph_timed else
                                 'the timed steps') + ': hipEvent pairs on the launch stream around every %d-th conv '
                                'launch, phase rotating with the step' % period) + (
                '; %d lanes: durations overlap (see DESIGN section 6)' % lanes_used if lanes_used > 1 else '')
        roof['launch_sampling'] = 'every %d-th conv launch of the event legs timed (hipEvent pairs)' % period
        rec = {
            'metric': 'candidate-latent evals/sec (fwd+loss+bwd), BigGAN-256 pop=18',
            'value': round(evals / elapsed, 3),
            'unit': 'evals/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True,
            'scaling': 'strong',
            'vs_baseline': None,
            'dtype': ('f32 (fp32 tensors everywhere; convolutions >= 32x32 and the self-attention multiply on '
                      'the 16-bit MFMA pipe with fp32-grade operand splits and fp32 accumulate: the 16x16 '
                      'Winograd, direct and sub-pixel 3x3 kernels and the 1x1 kernels (4x4 layers included) on '
                      'power-of-two scaled operands in 2 fp16 pieces, 3 products, the fused attention too; the 3-channel '
                      'image convs and the launches without handed-over maxima on 3 bf16 pieces, 6 products; dense '
                      'layers, reductions and elementwise work exact fp32)'
                      if bf3 else 'f32'),
            'data': 'synthetic',
            'config': {
                'workload': 'BigGAN-deep-256 BasinCMA inner step (pycma popsize 18, z in R^128): '
                            'Clamp hook -> generator fwd -> weighted L1 + 10*LPIPS-%s -> '
                            'bwd to (z,c) -> Adam; 256x256 synthetic target'
                            % ('VGG16' if args.lpips_net == 'vgg' else 'AlexNet'),
                'population': POP,
                'max_batch_size': MAX_BATCH,
                'exec_batch_size': args.exec_batch,
                'lanes': lanes_used,
                'lanes_note': 'reference chunks of 9 on %d HIP streams, one set of workspaces each; same bits as one stream (tests/test_shard_bits_gpu.py)' % lanes_used if lanes_used > 1 else 'one stream',
                'conv3x3_arithmetic': 'fp16x2 (16x16 Winograd kernel >= 128 channels; direct / sub-pixel kernel for 64-channel, up-sampling and small layers; 1x1 kernels; fused attention) + bf16x3 (3-channel image convs)' if bf3 else 'f32',
                'lpips_net': args.lpips_net,
                'parallelism': 'population sharded over %d rank(s)' % world,
                'rccl_ranks': dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
                'backend': (dist.get_backend() if (world > 1 and dist.is_initialized()) else None),
                'loss_gather': 'one all-gather per generation (lazy); per step only when log=True',
                'hip_graph_replay': bool(graph_replayed),
                'timed_steps_run_as': ('one HIP graph per step, replayed (two branches = the two lanes): the default of '
                                       'optimize() for this configuration' if graph_replayed else
                                       'eager launches' + (' with hipEvent pairs inside' if not graph_timed else '')),
                'gflop_per_eval_basis': gflop_eval,
                'end_to_end_tflops': round(gflop_eval * evals / elapsed / 1e3, 2),
                'fwd_only_rescore_evals_per_s': round(rescore_rate, 1),
                'last_losses_min_max': [round(min(last_loss), 5), round(max(last_loss), 5)],
                'last_losses': [round(x, 6) for x in last_loss],
            },
            'roofline': roof,
            'telemetry': telemetry.summary(),
        }
        if world == 1 and bf3 and not args.no_fp32_leg:
            # the same steps with every conv on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32),
            # reported beside `value` so the arithmetic choice is visible in one line
            rec['config']['exact_fp32_mfma_evals_per_s'] = round(exact_fp32_leg(dev, args), 1)
        if world == 1 and not args.no_extra:
            try:
                rec['config']['extra'] = extra_configs(dev)
            except Exception as e:          # the headline must not be lost to a side measurement
                rec['config']['extra'] = {'error': repr(e)[:300]}
            try:
                rec['config']['extra'].update(full_schedule_and_default_loss(dev, 1e3 * elapsed / args.steps))
            except Exception as e:
                rec['config']['extra']['full_schedule_error'] = repr(e)[:300]
        if world == 1 and not args.no_cpu_baseline:
            rec['cpu_baseline'] = cpu_baseline(problem)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
