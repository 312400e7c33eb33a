export TMPDIR=/tmp
for L in product tools/micro/lib_dephase.so product tools/micro/lib_dephase.so; do
  if [ $L = product ]; then unset P2L_LIB_PATH; else export P2L_LIB_PATH=$L; fi
  echo "== $L"; BATCHES=9,18 timeout 200 python tools/bench_wino_blocks.py 2>/dev/null | grep -E "B=9|B=18" | awk '{print $1,$2,$3,$4,$5,$6}' | tr "\n" ";"; echo
done
