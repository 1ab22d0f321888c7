#!/bin/bash
# rocprofv3 kernel trace of the default bench (2 timed steps + 1 warm-up); usage: tools/prof_bench.sh <tag>
# writes gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv you want judged into profiles/
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -o runc -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_$1_bench.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_$1
