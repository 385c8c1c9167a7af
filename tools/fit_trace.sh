cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fprof; PYTHONPATH=$GRAFT_REPO_ROOT NSTREAMS=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/fprof -- python $GRAFT_REPO_ROOT/tools/two_streams.py > /tmp/f.log 2>&1
tail -1 /tmp/f.log
t=$(ls -t /tmp/fprof/*/*_kernel_trace.csv | head -1)
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $t 2 -v 2>&1 > $GRAFT_REPO_ROOT/gpurun_out/fit_timeline.txt
