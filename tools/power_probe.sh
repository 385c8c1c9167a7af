#!/bin/bash
# Socket power and shader clock while the fp64 GEMM runs back to back (K=512 vs K=8192): is the kernel power-capped?
R=${GRAFT_REPO_ROOT:-/root/repo}
for K in 512 8192; do
  iters=$((K == 512 ? 3000 : 200))
  python $R/tools/gemm_only.py 8192 $K 0 1.0 128 $iters > /tmp/g_$K.out 2>&1 &
  pid=$!
  sleep 2.5
  for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
    sleep 0.4
  done
  wait $pid
  echo "K=$K:" $(tail -1 /tmp/g_$K.out)
done
