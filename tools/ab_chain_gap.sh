#!/bin/bash
# A/B of the accumulation-chain gaps (common.h ALIVE_CHAIN_GAP): shipped library against tools/_ab/nogap.so (knn.hip, filter_small.hip
# built with -DALIVE_KNN8_NO_CHAIN_GAP / -DALIVE_FBS_NO_CHAIN_GAP, filter_mid.hip of round 4)
run() {
  python tools/bench_filter_mid.py 2>&1 | grep "^fused"
  python tools/bench_filter_small.py 2>&1 | grep "^C"
  python tools/bench_knn.py 384 450 1000000 3 biased 2>&1 | tail -1
}
for i in 1 2; do
  echo "=== shipped"; run
  echo "=== tools/_ab/nogap.so"; ALIVE_VC_LIB=tools/_ab/nogap.so run
done
