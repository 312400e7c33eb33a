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

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

POP = 18
MAX_BATCH = 9
FP32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6      # dense v_mfma_f32_32x32x16_bf16 = 16 x the fp32 MFMA rate
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
    Wv = S.lpips_vgg_weights(1) if lpips_net == 'vgg' else S.lpips_alex_weights(2)
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


def cpu_baseline(problem, max_seconds=40.0):
    """the oracle (CPU restatement, torch-CPU fp32, all host cores) timed on a
    bounded sample of the same workload: fwd + loss + bwd of a chunk of 2
    candidates (dgrad only; the reference would also compute weight gradients)."""
    from oracle import biggan_ref as R, lpips_ref as L
    W, Wv, c_default, target, weight = problem
    n = 2
    g = torch.Generator().manual_seed(2)
    torch.set_num_threads(_best_threads())

    def one():
        z = torch.fmod(torch.randn(n, 128, generator=g), 2.0).requires_grad_(True)
        c = c_default.unsqueeze(0).repeat(n, 1).requires_grad_(True)
        out = R.biggan_forward(W, z, c)
        loss = L.projection_loss(Wv, out, target.unsqueeze(0).repeat(n, 1, 1, 1),
                                 weight.unsqueeze(0).repeat(n, 1, 1, 1))
        loss.mean().backward()
        return float(loss.sum())
    t0 = time.perf_counter()
    one()                                   # warm-up (thread pools, oneDNN primitives)
    warm = time.perf_counter() - t0
    reps = 0
    t0 = time.perf_counter()
    while True:
        one()
        reps += 1
        el = time.perf_counter() - t0
        if el + warm > max_seconds or reps >= 3:
            break
    return {'value': round(n * reps / el, 4), 'unit': 'evals/s',
            'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d x (fwd+loss+bwd of a chunk of %d candidates, BigGAN-deep-256 + '
                      'L1+10*LPIPS-VGG16, torch-CPU fp32, dgrad only)' % (reps, n)}


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
    ap.add_argument('--prof-period', type=int, default=4,
                    help='time every N-th conv launch with hipEvents inside the timed steps '
                         '(1 = every launch; costs ~5 %% of the step)')
    ap.add_argument('--no-fp32-leg', action='store_true',
                    help='skip the extra exact-fp32-MFMA measurement reported in config')
    ap.add_argument('--lpips-net', default='vgg', choices=['vgg', 'alex'],
                    help="LPIPS network: 'vgg' = BASELINE.json's metric (default); 'alex' = the "
                         "reference's ProjectionLoss() default, reported as a side configuration")
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend: 'nccl' (= RCCL) or "
                    "'gloo' (test only: lets several ranks share one GPU)")
    ap.add_argument('--exec-batch', type=int, default=POP,
                    help='candidates pushed through the device per pass (semantic chunk '
                         'size stays max_batch_size=9); 9 = execute chunk by chunk like '
                         'the reference')
    args = ap.parse_args()

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

    for i in range(args.warmup):
        opt.step(variables, optimize=True, transform=(i == 0))
    sync()
    lib = N.lib()
    n_prof = 4096
    N.check(lib.p2l_prof_begin(n_prof), 'p2l_prof_begin')
    # hipEvents around EVERY conv launch cost the stream ~5 us each (measured: 5 % of the
    # step); time every PERIOD-th launch, rotating the phase with the step, so that each
    # launch of the step is timed once per PERIOD steps
    period = max(1, min(args.prof_period, args.steps))
    t0 = time.perf_counter()
    for i in range(args.steps):
        lib.p2l_prof_step(i, period)
        opt.step(variables, optimize=True)
    sync()
    elapsed = time.perf_counter() - t0
    flops = (C.c_double * 2)()
    ms = (C.c_double * 2)()
    cnt = (C.c_int32 * 2)()
    abytes = (C.c_double * 2)()
    N.check(lib.p2l_prof_end2(flops, ms, cnt, abytes), 'p2l_prof_end2')
    last_loss = [float(x) for x in opt.loss]
    # SURVEY 8(d) also asks for the fwd-only rate (the CMA re-score); outside the timed K steps
    sync()
    t1 = time.perf_counter()
    n_rescore = 3
    for _ in range(n_rescore):
        opt.step(variables, optimize=False)
    sync()
    rescore_rate = POP * n_rescore / (time.perf_counter() - t1)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        evals = POP * args.steps
        # generator fwd+dgrad 58.80 GMAC + LPIPS net fwd+dgrad (VGG16 40.08 | AlexNet 1.74 GMAC)
        gflop_eval = GFLOP_PER_EVAL if args.lpips_net == 'vgg' else 2 * (58.80 + 1.737)
        conv_tflops = flops[0] / (ms[0] * 1e-3) / 1e12 if ms[0] > 0 else 0.0
        conv1_tflops = flops[1] / (ms[1] * 1e-3) / 1e12 if ms[1] > 0 else 0.0
        bf3 = N.default_wfmt() == N.WFMT_BF16X3
        # PMC counters cannot be read from inside the timed process: `traffic` is the
        # committed result of the separate rocprofv3 --pmc passes over this same command
        # (tools/gpu_profile.sh -> tools/traffic_json.py), or null when absent
        traffic, traffic_src = None, None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                             'round1_traffic.json')
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get('hbm_bytes_per_launch')
            traffic_src = 'profiles/round1_traffic.json'
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
            'dtype': ('f32 (3x3 convs: fp32-equivalent 3-way bf16 operand split on the bf16 MFMA '
                      'pipe, 6 products, fp32 accumulate; everything else exact fp32)'
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
                'conv3x3_arithmetic': 'bf16x3' if bf3 else 'f32',
                'lpips_net': args.lpips_net,
                'parallelism': 'population sharded over %d rank(s)' % world,
                'gflop_per_eval_basis': gflop_eval,
                'end_to_end_tflops': round(gflop_eval * evals / elapsed / 1e3, 2),
                'fwd_only_rescore_evals_per_s': round(rescore_rate, 1),
                'last_losses_min_max': [round(min(last_loss), 5), round(max(last_loss), 5)],
                'last_losses': [round(x, 6) for x in last_loss],
            },
            'roofline': {
                'kernel': ('conv_mfma_kernel<TAPS=9,BF3> (3x3 implicit GEMM, 6 x '
                           'v_mfma_f32_32x32x16_bf16 per 16 channels on 3-way split fp32 operands)'
                           if bf3 else
                           'conv_mfma_kernel<TAPS=9> (3x3 implicit GEMM, v_mfma_f32_32x32x2_f32)'),
                'bound': 'mfma',
                'achieved': round(conv_tflops, 2),
                # bf16x3: 6 bf16 MFMA products per fp32 product -> the matrix-pipe ceiling in
                # algorithmic (fp32-equivalent) FLOP/s is the dense bf16 peak / 6
                'peak': round(BF16_MFMA_PEAK_TFLOPS / 6, 1) if bf3 else FP32_MFMA_PEAK_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': round(conv_tflops / (BF16_MFMA_PEAK_TFLOPS / 6 if bf3
                                             else FP32_MFMA_PEAK_TFLOPS), 4),
                'vs_fp32_mfma_peak': round(conv_tflops / FP32_MFMA_PEAK_TFLOPS, 4),
                'issued_bf16_mfma_tflops': round(6 * conv_tflops, 1) if bf3 else None,
                'traffic': traffic,
                'traffic_unit': 'HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)',
                'traffic_source': traffic_src,
                'algo_bytes_per_launch': round(abytes[0] / max(cnt[0], 1)),
                'launches': int(cnt[0]),
                'launch_sampling': 'every %d-th conv launch timed (hipEvent pairs), phase rotating '
                                   'with the step' % period,
                'avg_launch_ms': round(ms[0] / max(cnt[0], 1), 4),
                'algo_gflop_per_launch': round(flops[0] / max(cnt[0], 1) / 1e9, 3),
                'time_share_of_step': round(period * ms[0] * 1e-3 / elapsed, 4),
                'conv1x1': {'achieved': round(conv1_tflops, 2), 'launches': int(cnt[1]),
                            'time_share_of_step': round(period * ms[1] * 1e-3 / elapsed, 4)},
            },
        }
        if world == 1 and bf3 and not args.no_fp32_leg:
            # the same steps with every conv on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32),
            # reported beside `value` so the arithmetic choice is visible in one line
            rec['config']['exact_fp32_mfma_evals_per_s'] = round(exact_fp32_leg(dev, args), 1)
        if world == 1 and not args.no_cpu_baseline:
            rec['cpu_baseline'] = cpu_baseline(problem)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
