#!/bin/bash
# per-dispatch kernel trace of the kNN search at the bench shape: tools/prof_knn_trace.sh <tag>
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ktrace_$1 -o knn -- \
    python3 $GRAFT_REPO_ROOT/tools/bench_knn.py 384 450 1000000 2 biased > $GRAFT_REPO_ROOT/gpurun_out/ktrace_$1.log 2>&1
F=$(ls $GRAFT_REPO_ROOT/gpurun_out/ktrace_$1/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/ktrace_list.py $F > $GRAFT_REPO_ROOT/gpurun_out/ktrace_$1_list.txt
tail -14 $GRAFT_REPO_ROOT/gpurun_out/ktrace_$1_list.txt
