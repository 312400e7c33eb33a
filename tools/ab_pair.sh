mkdir -p gpurun_out/pair
python tools/bench_subpix.py > gpurun_out/pair/bench_subpix.txt 2>&1
for i in 1 2 3; do
  for c in c4 c5; do
    python tools/step_sg2_one.py $c 2>/dev/null | head -1 | sed 's/^/pair    /' >> gpurun_out/pair/steps.txt
    P2L_LIB_PATH=$GRAFT_REPO_ROOT/tools/micro/libp2l_hip_ab.so python tools/step_sg2_one.py $c 2>/dev/null | head -1 | sed 's/^/single  /' >> gpurun_out/pair/steps.txt
  done
done
cat gpurun_out/pair/bench_subpix.txt gpurun_out/pair/steps.txt
