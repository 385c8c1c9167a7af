# kernel-trace timeline of ONE single-stream fit at N = 8192: bash tools/fit_trace.sh [out-name] [option=value ...]
out=${1:-fit_timeline}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fprof; PYTHONPATH=$GRAFT_REPO_ROOT NSTREAMS=${NSTREAMS:-1} rocprofv3 --kernel-trace --output-format csv -d /tmp/fprof -- python $GRAFT_REPO_ROOT/tools/two_streams.py "$@" > /tmp/f.log 2>&1
tail -1 /tmp/f.log
t=$(ls -t /tmp/fprof/*/*_kernel_trace.csv | head -1)
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $t 2 -v 2>&1 > $GRAFT_REPO_ROOT/gpurun_out/$out.txt
