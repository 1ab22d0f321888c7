#!/bin/bash
# PMC passes over the network kernels (tools/run_nets_once.py): one small counter group per pass, --kernel-trace only, every
# pass under its own timeout (a pass that combined FETCH_SIZE with WRITE_SIZE aborted and then hung until the box limit).
# usage: tools/pmc_nets.sh <tag>  ->  gpurun_out/pmc_<tag>/pass<i>/run_counter_collection.csv ; profiles/nets_mfma_pmc.json was
# aggregated from pass 1 (MFMA busy cycles / GRBM_GUI_ACTIVE per kernel)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$1/pass$i -o run -- \
      python3 $GRAFT_REPO_ROOT/tools/run_nets_once.py 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$1.pass$i.log 2>&1 || echo "pass $i ($grp) failed or timed out"
done
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$1/*
