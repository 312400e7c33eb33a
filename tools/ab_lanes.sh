# lanes x parts per reference chunk on the bench problem: value and ms per step of the timed region (graph replay)
mkdir -p gpurun_out/lanes
F="--no-cpu-baseline --no-extra --no-alone --no-fp32-leg --no-telemetry --steps 20 --warmup 3"
run() { P2L_STREAMS=$1 P2L_LANE_SPLIT=$2 python bench.py $F 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('streams $1 parts per chunk $2: %.1f evals/s %.3f ms lanes %s' % (r['value'], r['ms_per_step'], r['config'].get('lanes')))"; }
for rep in 1 2; do
run 2 1; run 4 2; run 2 2; run 3 3; run 6 3; run 3 1
done | tee gpurun_out/lanes/lanes.txt
