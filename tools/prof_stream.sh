#!/bin/bash
# rocprofv3 kernel trace of the streaming loop (200 eager steps); usage: tools/prof_stream.sh <tag>
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -o runc -- \
    python3 $GRAFT_REPO_ROOT/tools/bench_stream.py 200 > $GRAFT_REPO_ROOT/gpurun_out/prof_$1.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_$1
