#!/bin/bash
# A/B build of libalive_vc.so with one source recompiled under extra flags:  tools/ab_build.sh <out.so> <file.hip> [flags...]
# (the other objects are the ones of the last `make`; run from anywhere)
set -e
out=$1; src=$2; shift 2
cd "$(dirname "$0")/../alive-vc_amd/csrc"
extra=""
{ [ "$src" = "filter_mid.hip" ] || [ "$src" = "filter_small.hip" ]; } && extra="-fno-slp-vectorize -DALIVE_FILTER_MID_NO_SLP -mllvm -amdgpu-sched-strategy=max-ilp"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $extra "$@" -c $src -o /tmp/ab_$$.o
objs=""
for f in error conv conv_split conv_skinny gemm_planes filter_small filter_mid filter_big filter_edge blocks oscillator audio knn networks; do
  if [ "$f.hip" = "$src" ]; then objs="$objs /tmp/ab_$$.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$OLDPWD/$out" 2>/dev/null || /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$out"
rm -f /tmp/ab_$$.o
