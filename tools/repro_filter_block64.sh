#!/bin/bash
# Reproducer of the accumulation-chain hazard in the fused 64-channel FilterBlock (DESIGN.md 3.2b', csrc/filter_mid.hip header):
# the SAME source without the wait states between two MFMA accumulation chains (-DALIVE_FB64_NO_CHAIN_GAP) returns a wrong last
# accumulator register for the tiles whose epilogue runs beside the next chain's MFMAs -- some of them differently run to run.
# Builds the variant next to the product library and runs the determinism stress and the reference fixtures' unit test on both.
#   usage (on the GPU box): bash tools/repro_filter_block64.sh [launches]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p tools/_repro
bash tools/ab_build.sh tools/_repro/libalive_vc_no_chain_gap.so filter_mid.hip -DALIVE_FB64_NO_CHAIN_GAP
echo "shipped kernel:"; python3 tools/stress_filter_block.py ${1:-20} 2>&1 | grep "C=64"
python3 -m pytest tests/test_gpu_ops.py -m gpu -q -k "fused_filter_block_small and 64" 2>&1 | tail -1
echo "no wait states between the chains:"; ALIVE_VC_LIB=$R/tools/_repro/libalive_vc_no_chain_gap.so python3 tools/stress_filter_block.py ${1:-20} 2>&1 | grep "C=64"
ALIVE_VC_LIB=$R/tools/_repro/libalive_vc_no_chain_gap.so python3 -m pytest tests/test_gpu_ops.py -m gpu -q -k "fused_filter_block_small and 64" 2>&1 | tail -1
