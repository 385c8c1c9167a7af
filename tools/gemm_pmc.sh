#!/bin/bash
# PMC passes over the stand-alone trailing-update GEMM (8192 x 8192 output) at K = 512 and K = 8192: what the k-loop waits for.
#   gpurun -- 'bash tools/gemm_pmc.sh'   ->  gpurun_out/gemm_pmc.txt   (separate --pmc passes, --kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gemm_pmc; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"; do
  for K in 512 8192; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python $R/tools/gemm_only.py 8192 $K 0 1.0 128 3 > /dev/null 2> $O/p$i.err
    f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
    echo "== K=$K: $set"
    python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "gemm_f64_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
except Exception as e:
    print("  (no data:", e, ")")
for k in acc: print("  %-36s %.6g per launch (%d launches)" % (k, acc[k] / n[k], n[k]))
PY
  done
done > $R/gpurun_out/gemm_pmc.txt 2>&1
