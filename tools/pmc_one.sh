#!/bin/bash
# SQ stall / instruction-mix counters of one script's kernels: tools/pmc_one.sh <tag> <script.py> [args...]
# one small counter group per pass (--kernel-trace only), every pass under its own timeout
# -> gpurun_out/pmc1_<tag>/pass<i>/run_counter_collection.csv ; aggregate with tools/pmc_one_agg.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1_$tag/pass$i -o run -- \
      python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc1_$tag.pass$i.log 2>&1 || echo "pass $i ($grp) failed or timed out"
done
