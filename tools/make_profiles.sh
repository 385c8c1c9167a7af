#!/bin/bash
# Regenerate profiles/<round>_* on the GPU box:  gpurun -- 'bash tools/make_profiles.sh <git-head> [r06]'
# (kernel-trace stats and PMC counters in SEPARATE rocprofv3 runs; PMC runs use --kernel-trace only)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
HEAD=${1:-unknown}
RT=${2:-r06}
O=$R/gpurun_out/prof_$RT
rm -rf $O; mkdir -p $O/final
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
kstats() { find $1 -name "*kernel_stats.csv" | head -1; }
ccsv() { find $1 -name "*counter_collection.csv" | head -1; }
# 0. PMC traffic of gemm_f64 FIRST (separate passes, single stream): bench.py quotes the newest profiles/r*_gemm_f64_hbm_traffic.json whose
#    kernel fingerprint matches, so the file has to exist before the bench line of this set is taken; then a pause (the first process
#    after a counter pass starts at a fraction of the clock for a few seconds)
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$d -- python $R/bench.py --steps 3 --windows 1 --prof-steps 1 --streams 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_$d.err
  cp $(ccsv $O/pmc/$d) $O/final/${RT}_pmc_${d}_counter_collection.csv
done
python $R/tools/pmc_traffic.py $O/pmc $O/final/${RT}_gemm_f64_hbm_traffic.json $HEAD > /dev/null 2>> $O/pmc.err
cp $O/final/${RT}_gemm_f64_hbm_traffic.json $R/profiles/${RT}_gemm_f64_hbm_traffic.json
sleep 30; python $R/tools/two_streams.py > /dev/null 2>&1
# 1. the bench line itself (with cpu_baseline, extras)
timeout 1200 python $R/bench.py > $O/bench.json 2> $O/bench.err
grep '^{' $O/bench.json > $O/final/${RT}_bench.json
# 2a. kernel-trace stats of the SINGLE-STREAM command: the run whose per-kernel averages reproduce roofline.frac
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python $R/bench.py --streams 1 --no-cpu-baseline --no-extras > $O/s1.json 2> $O/stats1.err
grep '^{' $O/s1.json > $O/final/${RT}_bench_streams1_under_rocprof.json; cp $(kstats $O/stats1) $O/final/${RT}_bench_streams1_kernel_stats.csv
# 2b. the same for the timed (two fit streams) configuration: kernel time sums overlap there
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -- python $R/bench.py --no-cpu-baseline --no-extras > $O/s2.json 2> $O/stats2.err
grep '^{' $O/s2.json > $O/final/${RT}_bench_streams2_under_rocprof.json; cp $(kstats $O/stats2) $O/final/${RT}_bench_streams2_kernel_stats.csv
# (un-profiled timing runs come BEFORE the --pmc passes: right after a counter pass the next process starts at a fraction of the
#  clock for a few seconds -- measured: the N = 8192 fit of tools/sharded_time.py 10x slow directly behind xcd_decision.py)
# 2c. the sharded fit at world 1 against the single-GPU fit
timeout 900 python $R/tools/sharded_time.py 8192 16384 32768 65536 2> $O/sharded.err | grep "N=" > $O/final/${RT}_sharded_fit_world1.txt
# 2d. single- and two-stream rates of the raw C-ABI loop
NSTREAMS=1,2 timeout 300 python $R/tools/two_streams.py 2> /dev/null | grep fits > $O/final/${RT}_two_streams.txt
# 4. cfg 3 (SEard N=16384 d=64) and cfg 5 (EP N=4096 d=32): per-kernel stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3 -- python $R/tools/cfg3_time.py > $O/final/${RT}_cfg3_time.txt 2> $O/cfg3.err
cp $(kstats $O/cfg3) $O/final/${RT}_cfg3_seard_N16384_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5 -- python $R/tools/ep_time.py > $O/final/${RT}_cfg5_time.txt 2> $O/cfg5.err
cp $(kstats $O/cfg5) $O/final/${RT}_cfg5_ep_N4096_kernel_stats.csv
# 5. kernel assembly at N=16384 (RBF d=16, SEard d=64; full symmetric output and the fused factor form): stats + WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/asm -- python $R/tools/gpu_probe.py asm > $O/final/${RT}_assembly_probe.txt 2> $O/asm.err
cp $(kstats $O/asm) $O/final/${RT}_assembly_kernel_stats.csv
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/asm_w -- python $R/tools/gpu_probe.py asm > /dev/null 2> $O/asm_w.err
cp $(ccsv $O/asm_w) $O/final/${RT}_assembly_pmc_write_counter_collection.csv
# 6. the probes behind EXPERIMENTS.md (round 4): stand-alone GEMM rate by K, round quantisation, narrow outputs
( for k in 512 1024 2048 8192; do python $R/tools/gemm_only.py 8192 $k 0 1.0 128 10; done; python $R/tools/quant_probe.py; python $R/tools/narrow_probe.py ) > $O/final/${RT}_gemm_probes.txt 2> $O/probes.err
# 6b. (round 4, last third) the clock ramp, the bulk kernel from inside (stand-alone and inside a loop of fits), its PMC passes at
#     K = 512 / 8192, and what the assembly's stores cost by themselves
( python $R/tools/gemm_trace.py 8192 512 0 0 5; python $R/tools/gemm_trace.py 8192 512 0 0 100; python $R/tools/gemm_trace.py 8192 8192 0 0 10;
  python $R/tools/gemm_trace.py 2048 512 0 0 100; python $R/tools/fit_clock.py 1 12; python $R/tools/fit_clock.py 2 12;
  python $R/tools/two_streams.py pair_launch=0; python $R/tools/two_streams.py pair_launch=1 ) > $O/final/${RT}_gemm_inside.txt 2> $O/inside.err
bash $R/tools/gemm_pmc.sh > /dev/null 2>&1; cp $R/gpurun_out/gemm_pmc.txt $O/final/${RT}_gemm_pmc.txt
( hipcc --offload-arch=gfx950 -O3 -o /tmp/store_roof $R/tools/store_roof.hip && /tmp/store_roof 16384;
  python $R/tools/first_call.py 16384:10 16384:10 16384:10 16384:100 16384:100 ) > $O/final/${RT}_store_roof.txt 2> $O/store_roof.err
# 7. timeline of one single-stream fit: sched 0 (rounds 2-4), sched 1 (critical path on the panel stream), sched 2 (the default for a lone chain)
bash $R/tools/fit_trace.sh fit_timeline sched=0 > /dev/null 2>&1; cp $R/gpurun_out/fit_timeline.txt $O/final/${RT}_fit_timeline.txt
bash $R/tools/fit_trace.sh fit_timeline_sched2_s_pan sched=2 > /dev/null 2>&1; cp $R/gpurun_out/fit_timeline_sched2_s_pan.txt $O/final/${RT}_fit_timeline_sched2_s_pan.txt
# 8. round 5: the Gram-form assembly kernels (general / restructured / four workgroups per CU), sched 0 / 1 alternated, the leaf's phases
( python $R/tools/gram_probe.py -- "gram_fast=0,gram_grid=2048" "gram_fast=1,gram_grid=2048" "gram_fast=2,gram_grid=8192" "gram_fast=2,gram_grid=32768";
  python $R/tools/gram_probe.py d=32 -- "gram_fast=0,gram_grid=2048" "gram_fast=2,gram_grid=32768" ) > $O/final/${RT}_gram_probe.txt 2> $O/gram.err
( python $R/tools/ab_options.py N=8192 STEPS=40 ROUNDS=3 -- "sched=2" "sched=0" "trsm_lean=1" "trsm_lean=2" "publish=0"; python $R/tools/ab_options.py N=4096 STEPS=40 ROUNDS=3 -- "sched=2" "trsm_lean=1" "publish=0"; python $R/tools/gpu_probe.py leaf ) > $O/final/${RT}_option_ab_and_leaf_ticks.txt 2> $O/ab.err
( python $R/tools/predict_diag.py ) > $O/final/${RT}_predict_diag.txt 2> $O/predict.err
# 9. round 6, second half: the 128 x 64 tile and the folded kernel for clipped triangles; predict's two forms, the first predict after a
#    fit, and the same call with the BLAS pool left at one thread per visible core (the "hot chip" of rounds 3-5)
( python $R/tools/tur_probe.py; python $R/tools/_gemm_clip.py 3; echo "folded (gemm_f64_fold_kernel):"; PGP_TEST_GEMM_FOLD=1 python $R/tools/_gemm_clip.py 3 | sed -n 3,5p;
  python $R/tools/ab_options.py N=8192 STEPS=40 ROUNDS=3 -- "tur_tile=128" "tur_tile=1264"; python $R/tools/ab_options.py N=4096 STEPS=60 ROUNDS=3 -- "tur_tile=128" "tur_tile=1264" ) > $O/final/${RT}_tile_128x64_and_fold.txt 2> $O/tile.err
( python $R/tools/_pred_q.py; python $R/tools/_pred_first.py; echo "pools capped at the CPU quota (default):"; python $R/tools/_pred_t.py;
  echo "PYGPS_AMD_KEEP_THREADS=1:"; PYGPS_AMD_KEEP_THREADS=1 python $R/tools/_pred_t.py ) > $O/final/${RT}_predict_forms_and_host_threads.txt 2> $O/predict2.err
( REPS=20 python $R/tools/ep_kfold_diag.py 2>&1 | grep -E "rep|probe|gave" ) > $O/final/${RT}_ep_two_fit_streams_soak.txt
ls -la $O/final
