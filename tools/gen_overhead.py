"""Where the per-generation overhead of a BasinCMA run goes (bench.py config.extra.full_basincma_30x30_300:
38 ms per generation on top of the inner steps): the phases of search_loop.run_generations timed one by one
on the bench problem, GPU drained around each."""
import contextlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
opt, vm, _ = bench.build_problem(dev, exec_batch_size=int(os.environ.get("P2L_TOOL_EXEC", bench.MAX_BATCH)))
with contextlib.redirect_stdout(sys.stderr):
    opt.setup_cma(vm)
    opt.optimize(meta_steps=1, grad_steps=2, last_grad_steps=2)      # warm-up
    opt.setup_cma(vm)


def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t) * 1e3


acc = {}
for g in range(6):
    with torch.no_grad():
        variables, t_init = timed(lambda: vm.initialize(num_samples=opt.num_samples))
        _, t_draw = timed(lambda: opt.sampler.draw(variables, opt.shard))
    _, t_first = timed(lambda: opt.step(variables, optimize=True, transform=True))
    _, t_second = timed(lambda: opt.step(variables, optimize=True))
    _, t_steps = timed(lambda: [opt.step(variables, optimize=True) for _ in range(10)])
    with torch.no_grad():
        told, t_rescore = timed(lambda: opt.losses_for_tell(variables))
    _, t_tell = timed(lambda: opt.sampler.report(told))
    for k, v in (('initialize', t_init), ('draw (ask + 18 uploads)', t_draw), ('first step', t_first), ('second step', t_second),
                 ('steady step', t_steps / 10), ('re-score (+ host copy)', t_rescore), ('tell', t_tell)):
        acc.setdefault(k, []).append(v)
for k, v in acc.items():
    print('%-28s %7.2f ms  (median of %d generations; first %.2f)' % (k, float(np.median(v[1:])), len(v) - 1, v[0]))
s = {k: float(np.median(v[1:])) for k, v in acc.items()}
print('per generation on top of its steps: %.1f ms' % (s['initialize'] + s['draw (ask + 18 uploads)'] + (s['first step'] - s['steady step']) +
                                                      (s['second step'] - s['steady step']) + s['re-score (+ host copy)'] + s['tell']))
