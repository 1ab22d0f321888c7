#!/bin/bash
# rocprofv3 kernel trace of the default bench: 7 passes over the networks (1 warm-up, 2 timed, 2 PCIe-inclusive, 2 context-trimmed)
# and 8 scoring launches (those 7, the trimmed ones smaller, + 1 on uncorrelated frames); usage: tools/prof_bench.sh <tag>
# per-step table: python tools/kstats.py gpurun_out/prof_<tag>/runc_kernel_stats.csv 7
# writes gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv you want judged into profiles/
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -o runc -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_$1_bench.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_$1
