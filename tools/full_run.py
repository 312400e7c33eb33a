"""wall clock of a shortened BasinCMA run (config 3: BigGAN-256, pop 18) against the sum of its
inner steps: what the CMA generations cost on top (ask / fresh variables + Adam / re-score /
tell)."""
import contextlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench

dev = torch.device('cuda:0')
META, GRAD, LAST = 4, 30, 30
torch.manual_seed(0)
opt, vm, _ = bench.build_problem(dev, exec_batch_size=bench.POP)
opt.show_iter = 10 ** 9
with contextlib.redirect_stdout(sys.stderr):
    opt.optimize(meta_steps=1, grad_steps=3, last_grad_steps=3)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.optimize(meta_steps=META, grad_steps=GRAD, last_grad_steps=LAST)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    # inner steps alone
    opt.setup_cma(vm)
    variables = opt.cma_init(vm)
    for i in range(3):
        opt.step(variables, optimize=True, transform=(i == 0))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(30):
        opt.step(variables, optimize=True)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t1) / 30
n_steps = META * GRAD + LAST
print('BasinCMA %d generations x %d steps + %d: %.3f s wall; %d inner steps x %.2f ms = %.3f s; '
      'per generation on top: %.1f ms (%.1f %% of the run); %.0f evals/s over the whole run' % (
          META, GRAD, LAST, total, n_steps, step * 1e3, n_steps * step,
          (total - n_steps * step) / META * 1e3, 100 * (total - n_steps * step) / total,
          bench.POP * n_steps / total))
