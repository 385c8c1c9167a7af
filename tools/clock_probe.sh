#!/bin/bash
# Shader clock and socket power while (a) the K=512 fp64 GEMM runs back to back, (b) two fit streams, (c) one fit stream run.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
sample() { for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done; }
python tools/gemm_only.py 8192 512 0 1.0 128 4000 > /tmp/g.out 2>&1 &
pid=$!; sleep 3; echo "== GEMM K=512 back to back"; sample; wait $pid; tail -1 /tmp/g.out
for S in 2 1; do
  NSTREAMS=$S python tools/two_streams_long.py > /tmp/t.out 2>&1 &
  pid=$!; sleep 5; echo "== $S fit stream(s)"; sample; wait $pid; tail -1 /tmp/t.out
done
