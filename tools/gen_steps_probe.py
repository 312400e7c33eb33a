"""Host time of every inner step of a BasinCMA generation WITHOUT draining the GPU between steps (the real
loop), and the wall time of whole generations: where a generation loses time that tools/gen_overhead.py
(GPU drained around every phase) does not see."""
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
G, S = 5, 30
vm.reuse_buffers = os.environ.get('P2L_TOOL_REUSE', '1') == '1'      # (what run_generations does in graph mode)
if os.environ.get('P2L_TOOL_GCOFF'):
    import gc
    gc.collect(); gc.disable()
if os.environ.get('P2L_TOOL_GCFREEZE'):
    import gc
    gc.collect(); gc.freeze()
host = np.zeros((G, S)); walls = []; phases = []; segs = []
G = int(os.environ.get('P2L_TOOL_GENS', G)); host = np.zeros((G, S))
for g in range(G):
    seg0 = torch.cuda.memory_stats()['segment.all.allocated']
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        variables = vm.initialize(num_samples=opt.num_samples)
        opt.sampler.draw(variables, opt.shard)
    t1 = time.perf_counter()
    prof = None
    if os.environ.get('P2L_TOOL_PROFILE_HICCUP'):
        import cProfile, pstats
        prof = cProfile.Profile(); prof.enable()
    for i in range(S):
        t = time.perf_counter()
        opt.step(variables, optimize=True, transform=(i == 0))
        host[g, i] = (time.perf_counter() - t) * 1e3
        if prof is not None and i == 4:
            prof.disable()
            if host[g, :5].max() > 40 and g > 0:
                print('generation %d, steps 0-4: %s' % (g, np.round(host[g, :5], 1)))
                pstats.Stats(prof).sort_stats('tottime').print_stats(12)
                os.environ.pop('P2L_TOOL_PROFILE_HICCUP')
            prof = None
    t2 = time.perf_counter()
    with torch.no_grad():
        told = opt.losses_for_tell(variables)
    t3 = time.perf_counter()
    opt.sampler.report(told)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    walls.append((t4 - t0) * 1e3)
    segs.append(torch.cuda.memory_stats()['segment.all.allocated'] - seg0)
    phases.append([(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3])
print('wall per generation of %d steps: %s ms  (= %.2f ms per step)' % (S, np.round(walls, 1), np.median(walls[1:]) / S))
print('init+draw | %d steps (host) | re-score incl. wait for the GPU | tell : %s' % (S, np.round(np.median(np.array(phases)[1:], 0), 1)))
print('new allocator segments (hipMalloc) per generation:', segs)
if os.environ.get('P2L_TOOL_RAW'):
    for g in range(G):
        print(g, round(walls[g], 1), np.round(phases[g], 1), np.round(host[g], 1).tolist())
print('host ms of step i (median over generations):')
print(np.round(np.median(host[1:], 0), 1))
if os.environ.get('P2L_TOOL_CPROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    with torch.no_grad():
        variables = vm.initialize(num_samples=opt.num_samples)
        opt.sampler.draw(variables, opt.shard)
    opt.step(variables, optimize=True, transform=True)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
