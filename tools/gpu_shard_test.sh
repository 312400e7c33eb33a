#!/bin/bash
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg 2>/dev/null > gpurun_out/shard1.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo 2> gpurun_out/shard2.err > gpurun_out/shard2.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 3 --warmup 1 --backend gloo 2> gpurun_out/shard4.err > gpurun_out/shard4.json
tail -3 gpurun_out/shard2.err
python - <<'PY'
import json
r=[json.loads([l for l in open('gpurun_out/shard%d.json'%n).read().splitlines() if l.startswith('{"metric"')][-1]) for n in (1,2,4)]
import numpy as np
l=[np.array(x['config']['last_losses']) for x in r]
print('evals/s', [x['value'] for x in r])
print('max |loss(1 rank) - loss(2 ranks)|', np.abs(l[0]-l[1]).max(), ' (4 ranks)', np.abs(l[0]-l[2]).max())
print('rank order equal:', np.array_equal(np.argsort(l[0]), np.argsort(l[1])), np.array_equal(np.argsort(l[0]), np.argsort(l[2])))
PY
