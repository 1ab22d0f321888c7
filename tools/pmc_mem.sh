#!/bin/bash
# vector-memory / L1 / L2 counters of one script's kernels: tools/pmc_mem.sh <tag> <script.py> [args...]
# one small group per pass (--kernel-trace only); aggregate with tools/pmc_one_agg.py gpurun_out/pmc1_<tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN2_sum TCP_GATE_EN1_sum" "TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1_$tag/pass$i -o run -- \
      python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc1_$tag.pass$i.log 2>&1 || echo "pass $i ($grp) failed or timed out"
done
