#!/bin/bash
# instruction-cache / fetch counters of one script's kernels: tools/pmc_icache.sh <tag> <script.py> [args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1_$tag/pass$i -o run -- \
      python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc1_$tag.pass$i.log 2>&1 || echo "pass $i ($grp) failed or timed out"
done
