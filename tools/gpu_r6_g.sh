#!/bin/bash
mkdir -p gpurun_out/r6g
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g
cd $R
python tools/step_sg2_one.py c5 $O/c5.txt 2>/dev/null | head -1
python - <<'PY' 2>/dev/null
import sys, json, torch, contextlib
sys.path.insert(0, '.')
import bench
with contextlib.redirect_stdout(sys.stderr):
    pass
dev = torch.device('cuda:0')
out = bench.extra_configs(dev)
print({k: (v['evals_per_s'], v['ms_per_step'], v.get('hip_graph_replay')) for k, v in out.items()})
PY
