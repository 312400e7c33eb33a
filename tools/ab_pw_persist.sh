# full-tile pointwise kernel: one tile per block (product) vs persistent blocks (-DP2L_PW_PERSIST_GRID=1024 build)
mkdir -p gpurun_out/pwp
for rep in 1 2; do
python tools/bench_pw_h2.py 2>/dev/null
P2L_LIB_PATH=$GRAFT_REPO_ROOT/tools/micro/libp2l_hip_ab.so python tools/bench_pw_h2.py 2>/dev/null
done | tee gpurun_out/pwp/bench_pw.txt
