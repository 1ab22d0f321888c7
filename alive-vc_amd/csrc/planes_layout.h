// The k-blocked layout of plane-packed bf16 operands (include/alive_vc.h, "plane-packed operands").
//
// An operand of R rows (padded) and K reduction indices (a multiple of 32), split into NP bf16 planes, is stored as
//   [plane][K / 32][R][32]      element (plane, row, k)  ->  ((plane * K/32 + k/32) * R + row) * 32 + k % 32
// The 64-byte k-segments of consecutive rows are consecutive in memory: the K-step of a tile (rows r0 .. r0 + 127, one k-block) is
// one contiguous run, so the LDS-DMA pieces of the MFMA kernels (16 rows x 64 B) and the fragment-shaped reads of the kernels
// that take their weights straight from L2 touch whole 128-B lines, and every line is asked for once.
#pragma once
#include <stddef.h>
#include <stdint.h>

constexpr int PLANES_KB = 32;                 // k per block

__host__ __device__ __forceinline__ size_t planes_at(int plane, int64_t row, int k, int64_t R, int K) {
    return (((size_t)plane * (size_t)(K / PLANES_KB) + (size_t)(k / PLANES_KB)) * (size_t)R + (size_t)row) * PLANES_KB + (size_t)(k % PLANES_KB);
}
