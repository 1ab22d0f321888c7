#!/bin/bash
# rocprofv3 kernel trace of the headline step only (no secondary legs): 1 warm-up + 2 timed steps = 3 passes over the networks,
# 3 scoring launches.  ALIVE_STREAMS=1 so that kernel durations add up to the step.     usage: tools/prof_bench.sh <tag>
# per-step table: python tools/kstats.py gpurun_out/prof_<tag>/runc_kernel_stats.csv 3
# writes gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv you want judged into profiles/
cd /tmp && export TMPDIR=/tmp
ALIVE_STREAMS=${ALIVE_STREAMS:-1} rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -o runc -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --legs none --cpu-seconds 0 --no-nets-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_$1_bench.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_$1
