#!/bin/bash
# PMC passes over one launch of the first-stage scoring kernel (fp6 by default; ALIVE_KNN_PREFILTER=fp8 for the fp8 kernel) (tools/bench_knn.py 384 450 1000000 1 biased): one counter group per pass,
# --kernel-trace only, every pass under its own timeout.  usage: tools/pmc_knn8.sh <tag> -> gpurun_out/pmc8_<tag>/pass<i>/
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc8_$1/pass$i -o run -- \
      python3 $GRAFT_REPO_ROOT/tools/bench_knn.py 384 450 1000000 1 biased > $GRAFT_REPO_ROOT/gpurun_out/pmc8_$1.pass$i.log 2>&1 || echo "pass $i ($grp) failed or timed out"
done
ls $GRAFT_REPO_ROOT/gpurun_out/pmc8_$1/*
