#!/bin/bash
# The sharded path on whatever this box has.  On a node with N >= 2 GPUs every rank gets its own GPU and
# the collectives run over RCCL (backend nccl): the SCALE lines of 1 / 2 / 4 / 8 GPUs with
# config.rccl_ranks = N, tests/test_parallel_nccl_gpu.py, and the bit-identity of the losses across rank
# counts (tools/shard_bits.py).  On a 1-GPU box the ranks share the GPU over gloo (what the -m gpu suite
# does): same code path, no scaling figure.
#   gpurun -- 'bash tools/gpu_shard_test.sh'            results under gpurun_out/shard/
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.."
OUT=gpurun_out/shard; mkdir -p $OUT
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
BACKEND=gloo; [ "$NGPU" -ge 2 ] && BACKEND=nccl
echo "GPUs on this box: $NGPU -> backend $BACKEND"
run() {  # run <world> <script> <args...>
  local n=$1; shift
  if [ "$n" = 1 ]; then python "$@"; else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29600 + n)) "$@" --backend $BACKEND
  fi
}
for n in 1 2 4 8; do
  if [ "$BACKEND" = nccl ] && [ "$n" -gt "$NGPU" ]; then continue; fi
  if [ "$BACKEND" = gloo ] && [ "$n" -gt 4 ]; then continue; fi      # (8 processes on one GPU: nothing to learn)
  if [ "$n" = 1 ]; then run 1 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-extra > $OUT/scale_$n.json 2> $OUT/scale_$n.err
  else run "$n" bench.py --gpus "$n" --steps 10 --warmup 3 > $OUT/scale_$n.json 2> $OUT/scale_$n.err; fi
  if [ "$n" = 1 ]; then run 1 tools/shard_bits.py > $OUT/bits_$n.json 2> $OUT/bits_$n.err
  else run "$n" tools/shard_bits.py > $OUT/bits_$n.json 2> $OUT/bits_$n.err; fi
done
[ "$NGPU" -ge 2 ] && python -m pytest tests/test_parallel_nccl_gpu.py -q -m gpu > $OUT/nccl_tests.log 2>&1
python - <<'PY'
import json, glob, os
out = 'gpurun_out/shard'
def line(path, key):
    try:
        return json.loads([l for l in open(path) if l.startswith('{"' + key)][-1])
    except Exception:
        return None
base = line(out + '/scale_1.json', 'metric')
bits1 = line(out + '/bits_1.json', 'world')
for n in (1, 2, 4, 8):
    r, b = line('%s/scale_%d.json' % (out, n), 'metric'), line('%s/bits_%d.json' % (out, n), 'world')
    if r is None:
        continue
    print('N=%d  backend %-4s rccl_ranks %s  %8.1f evals/s  %6.2f ms/step  x%.2f of one GPU | losses of 3 steps + re-score + final latents vs 1 rank: %s' % (
        n, r['config']['backend'], r['config']['rccl_ranks'], r['value'], r['ms_per_step'], r['value'] / base['value'],
        'n/a' if (b is None or bits1 is None) else
        ('BIT-IDENTICAL' if (b['steps'], b['rescore'], b['z_bits_sum']) == (bits1['steps'], bits1['rescore'], bits1['z_bits_sum']) else 'DIFFER')))
if os.path.exists(out + '/nccl_tests.log'):
    print(open(out + '/nccl_tests.log').read().strip().splitlines()[-1])
PY
