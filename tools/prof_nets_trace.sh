#!/bin/bash
# per-dispatch kernel trace of one pass over the networks (tools/run_nets_once.py): tools/prof_nets_trace.sh <tag>
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$1 -o nets -- \
    python3 $GRAFT_REPO_ROOT/tools/run_nets_once.py 2 > $GRAFT_REPO_ROOT/gpurun_out/trace_$1.log 2>&1
F=$(ls $GRAFT_REPO_ROOT/gpurun_out/trace_$1/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/ktrace_list.py $F > $GRAFT_REPO_ROOT/gpurun_out/trace_$1_list.txt
tail -3 $GRAFT_REPO_ROOT/gpurun_out/trace_$1_list.txt
