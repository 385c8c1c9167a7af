#!/bin/bash
# Regenerate profiles/r01_* on the GPU box:  gpurun -- 'bash tools/make_profiles.sh'
# (kernel-trace stats and PMC counters in SEPARATE rocprofv3 runs; PMC runs use --kernel-trace only)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the bench line itself (with cpu_baseline)
timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err
# 2. kernel-trace stats of the same command (no cpu baseline: host time only)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/stats.err
# 3. PMC traffic of gemm_f64 (separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$d -- python $R/bench.py --steps 3 --streams 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$d.err
done
# 4. assembly kernel counters at N=16384
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "WRITE_SIZE" "FETCH_SIZE"; do
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/asm/$(echo $set | cut -c1-14 | tr " " _) -- python $R/tools/asm_only.py > $O/asm_only.out 2>> $O/asm.err
done
ls -R $O | head -50
