"""forward-only re-score of the bench population, repeated: every repetition must give the same bits
(deterministic reductions everywhere).  Prints the number of distinct loss vectors."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
opt, vm, _ = bench.build_problem(dev)
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    opt.setup_cma(vm)
    variables = opt.cma_init(vm)
for j in range(2):
    opt.step(variables, optimize=True, transform=(j == 0))
res = []
for i in range(int(os.environ.get('REPS', 25))):
    _, l, _ = opt.step(variables, optimize=False)
    res.append(np.array(l, dtype=np.float64))
res = np.stack(res)
uniq = np.unique(res, axis=0)
diff = (res != res[0]).any(axis=0)
print(os.environ.get('P2L_LIB_PATH', 'product'), 'distinct loss vectors:', len(uniq), 'candidates that ever differ:', np.nonzero(diff)[0].tolist(),
      'max |d|: %.3g' % np.abs(res - res[0]).max())
