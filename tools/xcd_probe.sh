cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
for o in "xcd_order=1 xcd_min_tiles=256" "xcd_order=1 xcd_min_tiles=128"; do
  NSTREAMS=1,2 python $R/tools/two_streams.py $o 2>&1 | tail -2
  rm -rf /tmp/pm; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm -- python $R/tools/stage_times.py $o > /dev/null 2>&1
  python - "$o" <<PY
import csv,glob,sys
tot=n=0
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE" and "gemm_f64_kernel<128, 128, false, false, true, false>" in r["Kernel_Name"]:
            tot+=float(r["Counter_Value"]); n+=1
print(sys.argv[1], "FETCH per launch (x2, MB):", tot*2048/n/1e6, "launches", n)
PY
done
