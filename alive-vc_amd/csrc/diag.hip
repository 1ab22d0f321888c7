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

// block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (scale 2^0 on both operands): FMT 0 = fp8 e4m3 (8 registers per fragment),
// 2 = fp6 e2m3 (6), 4 = fp4 (4).  LDSA: the A fragment of every two MFMAs comes from LDS (32 / 24 / 16 bytes per lane).
typedef int v8i __attribute__((ext_vector_type(8)));
template <int FMT, bool LDSA>
__global__ __launch_bounds__(256, 1) void mfma_scale_rate_kernel(const int* __restrict__ rnd, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 1024 / 16; i += 256) ((u32x4*)lds)[i] = ((const u32x4*)rnd)[i] & 0xf7f7f7f7u;     // no e4m3 NaN
    __syncthreads();
    v8i fa[4], fb[8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[i][e] = rnd[((tid * 4 + i) * 8 + e) % 16384] & 0xf7f7f7f7;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) fb[i][e] = rnd[((tid * 8 + i + 2048) * 8 + e) % 16384] & 0xf7f7f7f7;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const unsigned char* base = lds + lane * 32;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (LDSA) {
                const unsigned char* q = base + (((it * 4 + i) * 2048) & 0xf800);
                const u32x4 lo = *(const u32x4*)q;
                fa[i][0] = lo[0]; fa[i][1] = lo[1]; fa[i][2] = lo[2]; fa[i][3] = lo[3];
                if (FMT == 0) { const u32x4 hi = *(const u32x4*)(q + 16); fa[i][4] = hi[0]; fa[i][5] = hi[1]; fa[i][6] = hi[2]; fa[i][7] = hi[3]; }
                if (FMT == 2) { const uint2 hi = *(const uint2*)(q + 16); fa[i][4] = hi.x; fa[i][5] = hi.y; }
            }
            acc[2 * i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[i], fb[2 * i], acc[2 * i], FMT, FMT, 0, 127, 0, 127);
            acc[2 * i + 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[i], fb[2 * i + 1], acc[2 * i + 1], FMT, FMT, 0, 127, 0, 127);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
    if (s == 12345.678f) sink[0] = s;
}

}  // namespace

// launches `blocks` blocks of the loop above; 8 * iters MFMAs per wave.  rnd: 64 KB of random bf16.
extern "C" int alive_debug_mfma_rate(const void* rnd, int blocks, int iters, int mode, float* sink, void* stream) {
    ALIVE_CHECK_ARG(rnd && sink && blocks > 0 && iters > 0, "alive_debug_mfma_rate: bad args");
    hipStream_t s = (hipStream_t)stream;
    const int* ri = (const int*)rnd;
    if (mode == 2) mfma_scale_rate_kernel<0, false><<<blocks, 256, 0, s>>>(ri, iters, sink);
    else if (mode == 3) mfma_scale_rate_kernel<2, false><<<blocks, 256, 0, s>>>(ri, iters, sink);
    else if (mode == 4) mfma_scale_rate_kernel<4, false><<<blocks, 256, 0, s>>>(ri, iters, sink);
    else if (mode == 5) mfma_scale_rate_kernel<0, true><<<blocks, 256, 0, s>>>(ri, iters, sink);
    else if (mode == 6) mfma_scale_rate_kernel<2, true><<<blocks, 256, 0, s>>>(ri, iters, sink);
    else mfma_rate_kernel<<<blocks, 256, 0, s>>>((const unsigned short*)rnd, iters, mode, sink);
    ALIVE_CHECK_LAUNCH("alive_debug_mfma_rate");
    return ALIVE_OK;
}
