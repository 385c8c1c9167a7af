cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/epprof; PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/epprof -- timeout 120 python $GRAFT_REPO_ROOT/tools/ep_time.py > /tmp/ep.log 2>&1
tail -3 /tmp/ep.log
f=$(ls -t /tmp/epprof/*/*_kernel_stats.csv | head -1)
head -16 $f | cut -c1-200
t=$(ls -t /tmp/epprof/*/*_kernel_trace.csv | head -1)
gzip -c $t > $GRAFT_REPO_ROOT/gpurun_out/ep_trace.csv.gz
PYTHONPATH=$GRAFT_REPO_ROOT python $GRAFT_REPO_ROOT/tools/ep_timeline.py $t > $GRAFT_REPO_ROOT/gpurun_out/ep_timeline.txt 2>&1
