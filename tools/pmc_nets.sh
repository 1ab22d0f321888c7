#!/bin/bash
# PMC passes over the network kernels (one counter group per pass, no tracing besides --kernel-trace); usage: tools/pmc_nets.sh <tag>
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$1/pass$i -o run -- \
      python3 $GRAFT_REPO_ROOT/tools/run_nets_once.py 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$1.pass$i.log 2>&1
done
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$1/*
