"""How much of the inner step is launch gaps?  Capture one step in a HIP graph
(torch.cuda.CUDAGraph) and compare replay with eager execution.  Timing probe only: Adam's
step counter is baked into the captured graph."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import contextlib
dev = torch.device('cuda:0')
for pop in (18, 3):
    bench.POP = pop
    torch.manual_seed(0)
    opt, vm, _ = bench.build_problem(dev, exec_batch_size=pop)
    with contextlib.redirect_stdout(sys.stderr):
        opt.num_samples = pop
        variables = vm.initialize(num_samples=pop)
    opt.track_variables = False
    opt.use_graph = False          # this probe captures the step itself
    for i in range(3):
        opt.step(variables, optimize=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        opt.step(variables, optimize=True)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        opt.step(variables, optimize=True)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        opt.step(variables, optimize=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / 10
    print('pop %2d: eager %.3f ms/step, graph replay %.3f ms/step (x%.3f)' % (pop, eager * 1e3, rep * 1e3, eager / rep))
