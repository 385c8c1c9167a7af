#!/bin/bash
# PMC passes over the stand-alone trailing-update-shaped GEMM (M=N=8192, K=512, beta=1): where do the waves wait?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_gemm
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$(echo $set | cut -c1-20 | tr " " _)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$d -- python $R/tools/gemm_only.py 8192 ${1:-512} 0 1.0 128 3 > $O/$d.out 2> $O/$d.err
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f64_kernel" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print("%-28s %16.0f per launch (n=%d)" % (k, v / n, n))
PY
