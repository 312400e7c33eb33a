# runtime environment knobs on the bench step (two lanes, graph replay) and on the 3-candidate step
mkdir -p gpurun_out/env
F="--no-cpu-baseline --no-extra --no-alone --no-fp32-leg --no-telemetry --steps 20 --warmup 3"
run() { env $1 python bench.py $F 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('%-40s %.1f evals/s %.3f ms' % ('$1', r['value'], r['ms_per_step']))"; }
for rep in 1 2; do
run A=0; run GPU_MAX_HW_QUEUES=2; run GPU_MAX_HW_QUEUES=8; run HIP_FORCE_DEV_KERNARG=0; run HIP_FORCE_DEV_KERNARG=1; run HSA_ENABLE_SDMA=0; run DEBUG_HIP_GRAPH_DOT_PRINT=0; run AMD_SERIALIZE_KERNEL=0; run HIP_LAUNCH_BLOCKING=0
done | tee gpurun_out/env/env.txt
