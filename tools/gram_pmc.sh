#!/bin/bash
# PMC passes over the Gram-form assembly (full symmetric SEard K, N = 16384, d = 64 and d = 32): where its waves' cycles go.
#   gpurun -- 'bash tools/gram_pmc.sh'   ->  gpurun_out/gram_pmc.txt   (separate --pmc passes, --kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gram_pmc; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_VMEM_WR" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "SQ_WAVES SQ_INST_LEVEL_VMEM"; do
  for D in 64 32; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python $R/tools/gram_probe.py d=$D ROUNDS=1 -- gram_fast=2 > /dev/null 2> $O/p$i.err
    f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
    echo "== d=$D: $set"
    python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "cov_gram_fast_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
except Exception as e:
    print("  (no data:", e, ")")
for k in acc: print("  %-36s %.6g per launch (%d launches)" % (k, acc[k] / n[k], n[k]))
PY
  done
done > $R/gpurun_out/gram_pmc.txt 2>&1
