#!/bin/bash
# Minimal reproducer of the code-generation defect in filter_block64_kernel (DESIGN.md 3.2b'): the SAME source with the two
# cross-term MFMAs on separate accumulators (-DALIVE_FB64_3CHAINS), compiled with the shipped flags (-fno-slp-vectorize, per-tile
# fence in place), gives run-to-run different results at 128 windows.  Builds the variant next to the product library and runs
# the determinism stress on both.   usage (on the GPU box): bash tools/repro_filter_block64.sh [launches]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/alive-vc_amd/csrc
mkdir -p $R/tools/_repro
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -DALIVE_FILTER_MID_NO_SLP \
    -DALIVE_FB64_3CHAINS -I. -I../../include -c filter_mid.hip -o $R/tools/_repro/filter_mid_3chains.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC error.o conv.o conv_split.o conv_skinny.o gemm_planes.o filter_small.o \
    $R/tools/_repro/filter_mid_3chains.o filter_edge.o blocks.o oscillator.o audio.o knn.o networks.o -o $R/tools/_repro/libalive_vc_3chains.so
cd $R
echo "shipped kernel:"; python3 tools/stress_filter_block.py ${1:-20} 2>&1 | grep "C=64"
echo "three accumulator chains:"; ALIVE_VC_LIB=$R/tools/_repro/libalive_vc_3chains.so python3 tools/stress_filter_block.py ${1:-20} 2>&1 | grep "C=64"
