#!/bin/bash
# Regenerate profiles/r02_* on the GPU box:  gpurun -- 'bash tools/make_profiles.sh'
# (kernel-trace stats and PMC counters in SEPARATE rocprofv3 runs; PMC runs use --kernel-trace only)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
rm -rf $O/stats1 $O/stats2 $O/pmc $O/final
mkdir -p $O $O/final
cd /tmp && export TMPDIR=/tmp
# 1. the bench line itself (with cpu_baseline, extras)
timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err
# 2a. kernel-trace stats of the SINGLE-STREAM command: this is the run whose per-kernel averages reproduce roofline.frac
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python $R/bench.py --streams 1 --no-cpu-baseline --no-extras > $O/bench_streams1_under_rocprof.json 2> $O/stats1.err
# 2b. the same for the timed (two fit streams) configuration: kernel time sums overlap there
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- python $R/bench.py --no-cpu-baseline --no-extras > $O/bench_streams2_under_rocprof.json 2> $O/stats2.err
# 3. PMC traffic of gemm_f64 (separate passes, single stream)
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$d -- python $R/bench.py --steps 3 --windows 1 --prof-steps 1 --streams 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$d.err
done
python $R/tools/pmc_traffic.py $O/pmc $O/r02_gemm_f64_hbm_traffic.json > /dev/null 2>> $O/pmc.err
# 4. the files profiles/ tracks, under their tracked names (copy gpurun_out/prof_r02/final/* to profiles/)
grep '^{' $O/bench.json > $O/final/r02_bench.json
grep '^{' $O/bench_streams1_under_rocprof.json > $O/final/r02_bench_streams1_under_rocprof.json
grep '^{' $O/bench_streams2_under_rocprof.json > $O/final/r02_bench_streams2_under_rocprof.json
cp $(ls -t $O/stats1/*/*_kernel_stats.csv | head -1) $O/final/r02_bench_streams1_kernel_stats.csv
cp $(ls -t $O/stats2/*/*_kernel_stats.csv | head -1) $O/final/r02_bench_streams2_kernel_stats.csv
cp $(ls -t $O/pmc/fetch/*/*_counter_collection.csv | head -1) $O/final/r02_pmc_fetch_counter_collection.csv
cp $(ls -t $O/pmc/write/*/*_counter_collection.csv | head -1) $O/final/r02_pmc_write_counter_collection.csv
cp $O/r02_gemm_f64_hbm_traffic.json $O/final/
ls -la $O/final
