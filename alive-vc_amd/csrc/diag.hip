// Diagnostic only (tools/mfma_rate.py): what bf16 MFMA rate this chip sustains, to put the scoring kernel's number in
// context.  One block per CU, one wave per SIMD (the scoring kernel's occupancy), v_mfma_f32_32x32x16_bf16 on RANDOM
// operands (the clock the chip holds depends on the data: MI355X_MICROARCH.md "DVFS give-back").
//   mode 0: operands in registers, 8 independent accumulators per wave
//   mode 1: the scoring kernel's operand traffic: one ds_read_b128 A fragment per two MFMAs, B fragments in registers
#include "common.h"

namespace {

__global__ __launch_bounds__(256, 1) void mfma_rate_kernel(const unsigned short* __restrict__ rnd, int iters, int mode, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 1024 / 16; i += 256) ((u32x4*)lds)[i] = ((const u32x4*)rnd)[i];
    __syncthreads();
    bf16x8 a[4], b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(rnd + ((tid * 4 + i) * 8) % 32768);
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = *(const bf16x8*)(rnd + ((tid * 8 + i + 4096) * 8) % 32768);
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    if (mode == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i], acc[i], 0, 0, 0);
        }
    } else {
        const unsigned char* base = lds + lane * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8 af = *(const bf16x8*)(base + (((it * 4 + i) * 1024) & 0xfc00));
                acc[2 * i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[2 * i], acc[2 * i], 0, 0, 0);
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[2 * i + 1], acc[2 * i + 1], 0, 0, 0);
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
    if (s == 12345.678f) sink[0] = s;                 // keeps the MFMAs alive
}

}  // namespace

// launches `blocks` blocks of the loop above; 8 * iters MFMAs per wave.  rnd: 64 KB of random bf16.
extern "C" int alive_debug_mfma_rate(const void* rnd, int blocks, int iters, int mode, float* sink, void* stream) {
    ALIVE_CHECK_ARG(rnd && sink && blocks > 0 && iters > 0, "alive_debug_mfma_rate: bad args");
    mfma_rate_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const unsigned short*)rnd, iters, mode, sink);
    ALIVE_CHECK_LAUNCH("alive_debug_mfma_rate");
    return ALIVE_OK;
}
