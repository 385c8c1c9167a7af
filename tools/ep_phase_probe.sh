# Phase ablation of ep_sites_lazy_kernel (timing only; results are wrong under ep_dbg != 0): average kernel duration
# from rocprofv3 kernel stats with the recurrence (1), the Gram phase (2) or the rows loop (4) removed.
cd /tmp && export TMPDIR=/tmp
for d in ${DBGS:-0 8 16 24}; do
rm -rf /tmp/epprof; PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/epprof -- timeout 120 python $GRAFT_REPO_ROOT/tools/ep_time.py ep_dbg=$d > /tmp/ep.log 2>&1
f=$(ls -t /tmp/epprof/*/*_kernel_stats.csv | head -1)
python - "$f" $d <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ep_sites_lazy" in r["Name"] or "ep_chain" in r["Name"] or "ep_prep" in r["Name"]: print("ep_dbg", sys.argv[2], "calls", r["Calls"], "avg us", float(r["AverageNs"])/1e3)
PY
done
