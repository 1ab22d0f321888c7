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

// Round 5: operand-feed forms that were priced before any kernel was rewritten (tools/mfma_rate.py modes 7 .. 10).
//   mfma_mixed_rate_kernel<BFMT, NB>: the fp8 scoring kernel's loop shape -- an fp8 A fragment (32 B per lane) from LDS feeds NB
//   MFMAs whose B operands sit in registers as fp8 (BFMT 0: 8 registers each) or fp6 e2m3 (BFMT 2: 6 registers); NB chains.
//   12 MFMAs per loop iteration.  NB = 2, fp8: what knn_score8_kernel does today (64 stationary frames per wave); NB = 3, fp6:
//   96 stationary frames per wave in 216 registers.
template <int BFMT, int NB>
__global__ __launch_bounds__(256, 1) void mfma_mixed_rate_kernel(const int* __restrict__ rnd, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 1024 / 16; i += 256) ((u32x4*)lds)[i] = ((const u32x4*)rnd)[i] & 0xf7f7f7f7u;     // no e4m3 NaN
    __syncthreads();
    constexpr int NF = 12 / NB;                       // A fragments per iteration
    v8i fb[12];
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) fb[i][e] = (e < (BFMT == 0 ? 8 : 6)) ? (rnd[((tid * 12 + i + 2048) * 8 + e) % 16384] & 0xf7f7f7f7) : 0;
    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const unsigned char* base = lds + lane * 32;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const unsigned char* q = base + (((it * NF + f) * 2048) & 0xf800);
            const u32x4 lo = *(const u32x4*)q, hi = *(const u32x4*)(q + 16);
            v8i fa;
            fa[0] = lo[0]; fa[1] = lo[1]; fa[2] = lo[2]; fa[3] = lo[3]; fa[4] = hi[0]; fa[5] = hi[1]; fa[6] = hi[2]; fa[7] = hi[3];
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb[f * NB + j], acc[j], 0, BFMT, 0, 127, 0, 127);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NB; ++i) s += acc[i][lane & 15];
    if (s == 12345.678f) sink[0] = s;
}

//   mfma_bf16_feed_kernel<NB>: the bf16 scoring kernel's loop shape with NB MFMAs per ds_read_b128 A fragment (NB = 2: 64 stationary
//   frames per wave, today; NB = 1: the 32-frames-per-wave form that would leave room for a second accumulator set).  8 MFMAs per iteration.
template <int NB>
__global__ __launch_bounds__(256, 1) void mfma_bf16_feed_kernel(const unsigned short* __restrict__ rnd, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 1024 / 16; i += 256) ((u32x4*)lds)[i] = ((const u32x4*)rnd)[i];
    __syncthreads();
    bf16x8 b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = *(const bf16x8*)(rnd + ((tid * 8 + i + 4096) * 8) % 32768);
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const unsigned char* base = lds + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < 8 / NB; ++f) {
            const bf16x8 af = *(const bf16x8*)(base + (((it * (8 / NB) + f) * 1024) & 0xfc00));
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[(f * NB + j) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[f * NB + j], acc[(f * NB + j) & 1], 0, 0, 0);
        }
    }
    float s = acc[0][lane & 15] + acc[1][lane & 15];
    if (s == 12345.678f) sink[0] = s;
}

// Round 4: what a vector instruction costs BESIDE an MFMA, one wave per SIMD (tools/mfma_filler.py).  One dependent chain of
// v_mfma_f32_32x32x16_bf16 (KIND 0, 1) or two alternating accumulators (KIND 2), NF v_fma_f32 after every MFMA: on NCH independent
// registers (KIND 0, 2: NCH = 8) or as ONE dependent chain (KIND 1).  All of it asm volatile, so the order is the source's.
// Returns shader cycles (s_memtime) per loop iteration of 8 MFMAs, per wave.
template <int NF, int KIND>
__global__ __launch_bounds__(256, 1) void mfma_filler_kernel(const unsigned short* __restrict__ rnd, int iters, long long* cycles, float* sink) {
    const int tid = threadIdx.x;
    bf16x8 a = *(const bf16x8*)(rnd + (tid * 8) % 32768), b = *(const bf16x8*)(rnd + (tid * 8 + 4096) % 32768);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)(tid + j) * 1e-3f;
    const float c1 = 0.999f, c2 = 1e-4f;
    __shared__ __attribute__((aligned(16))) unsigned char fl_lds[8192];
    u32x4 ld[4] = {};
    uint2 wr2 = make_uint2(tid, tid);
    if (KIND == 6 || KIND == 9 || KIND == 11) { ((u32x4*)fl_lds)[tid] = u32x4{1u, 2u, 3u, 4u}; __syncthreads(); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (KIND == 2 && (m & 1)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
            else if (KIND == 10) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "a"(a), "v"(b));      // A operand in an AGPR
            else if (KIND == 12) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));      // accumulator in VGPRs
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
            if (KIND == 11) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[m & 3]) : "v"(tid * 16));                          // one fragment read per MFMA besides the fillers
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float& xr = x[(m * NF + f) & 7];
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(c1), "v"(c2));
                else if (KIND == 3) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3fb5f0e3" : "+v"(xr) : "v"(c1));          // 32-bit literal
                else if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(xr));
                else if (KIND == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(xr) : "v"(c1));
                else if (KIND == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(m * NF + f) & 3]) : "v"(tid * 16));
                else if (KIND == 7) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(xr) : "a"(acc1[(m * NF + f) & 15]));
                else if (KIND == 8) asm volatile("v_fma_f32 %0, |%0|, %1, %2" : "+v"(xr) : "v"(c1), "v"(c2));        // VOP3 source modifier
                else if (KIND == 9) asm volatile("ds_write_b64 %0, %1" :: "v"(tid * 8), "v"(wr2));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xr) : "v"(c1), "v"(c2));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)(ld[0][0] + ld[1][1] + ld[2][2] + ld[3][3]);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    s += acc0[tid & 15] + acc1[tid & 15];
    if (s == 12345.678f) sink[0] = s;
    if ((tid & 63) == 0) cycles[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

}  // namespace

template <int KIND>
static void launch_filler(int nf, const unsigned short* rnd, int blocks, int iters, long long* cycles, float* sink, hipStream_t s) {
#define ALIVE_FILLER_CASE(N) case N: mfma_filler_kernel<N, KIND><<<blocks, 256, 0, s>>>(rnd, iters, cycles, sink); break;
    switch (nf) {
        ALIVE_FILLER_CASE(0) ALIVE_FILLER_CASE(1) ALIVE_FILLER_CASE(2) ALIVE_FILLER_CASE(3) ALIVE_FILLER_CASE(4) ALIVE_FILLER_CASE(5)
        ALIVE_FILLER_CASE(6) ALIVE_FILLER_CASE(7) ALIVE_FILLER_CASE(8) ALIVE_FILLER_CASE(10) ALIVE_FILLER_CASE(12) ALIVE_FILLER_CASE(16)
        default: break;
    }
#undef ALIVE_FILLER_CASE
}

// cycles[blocks * 4]: shader cycles of each wave for iters x 8 MFMAs with nf fillers behind every MFMA
extern "C" int alive_debug_mfma_filler(const void* rnd, int blocks, int iters, int nf, int kind, long long* cycles, float* sink, void* stream) {
    ALIVE_CHECK_ARG(rnd && sink && cycles && blocks > 0 && iters > 0, "alive_debug_mfma_filler: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0) launch_filler<0>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else if (kind == 1) launch_filler<1>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else if (kind == 2) launch_filler<2>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else if (kind == 3) launch_filler<3>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else if (kind == 4) launch_filler<4>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else if (kind == 5) launch_filler<5>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else if (kind == 6) launch_filler<6>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else if (kind == 7) launch_filler<7>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else if (kind == 8) launch_filler<8>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else if (kind == 9) launch_filler<9>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else if (kind == 10) launch_filler<10>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else if (kind == 11) launch_filler<11>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    else launch_filler<12>(nf, (const unsigned short*)rnd, blocks, iters, cycles, sink, s);
    ALIVE_CHECK_LAUNCH("alive_debug_mfma_filler");
    return ALIVE_OK;
}

// Round 6: does a SECOND wave on a SIMD double the vector issue rate?  (MI355X_MICROARCH.md: v_fma_f32 2 cycles per SIMD-32, "one wave
// alone: 4".)  The same loop as mfma_filler_kernel KIND 0 / a loop without MFMAs (MF = 0), launched with 256 threads (one wave per SIMD)
// or 512 (two): per-wave shader cycles per slot (slot = one MFMA or none + NF independent v_fma_f32).  tools/valu_pairs.py.
namespace {
template <int NF, int MF>
__global__ __launch_bounds__(512, 1) void valu_pair_kernel(const unsigned short* __restrict__ rnd, int iters, long long* cycles, float* sink) {
    const int tid = threadIdx.x;
    bf16x8 a = *(const bf16x8*)(rnd + (tid * 8) % 32768), b = *(const bf16x8*)(rnd + (tid * 8 + 4096) % 32768);
    f32x16 acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.0f;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)(tid + j) * 1e-3f;
    const float c1 = 0.999f, c2 = 1e-4f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float& xr = x[(m * NF + f) & 7];
                if (MF == 2 && (f & 7) == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(xr));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xr) : "v"(c1), "v"(c2));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    s += acc0[tid & 15];
    if (s == 12345.678f) sink[0] = s;
    if ((tid & 63) == 0) cycles[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}
}  // namespace

// cycles[blocks * 8]; threads = 256 (one wave per SIMD) or 512 (two); mf: 0 = no MFMA, 1 = one MFMA per slot, 2 = MFMA + every 8th filler a v_exp
extern "C" int alive_debug_valu_pairs(const void* rnd, int blocks, int threads, int iters, int nf, int mf, long long* cycles, float* sink, void* stream) {
    ALIVE_CHECK_ARG(rnd && sink && cycles && blocks > 0 && iters > 0 && (threads == 256 || threads == 512), "alive_debug_valu_pairs: bad args");
    hipStream_t s = (hipStream_t)stream;
    const unsigned short* r = (const unsigned short*)rnd;
#define ALIVE_VP_CASE(N, M) if (nf == N && mf == M) valu_pair_kernel<N, M><<<blocks, threads, 0, s>>>(r, iters, cycles, sink);
    ALIVE_VP_CASE(4, 0) ALIVE_VP_CASE(8, 0) ALIVE_VP_CASE(16, 0)
    ALIVE_VP_CASE(0, 1) ALIVE_VP_CASE(4, 1) ALIVE_VP_CASE(8, 1) ALIVE_VP_CASE(16, 1) ALIVE_VP_CASE(24, 1)
    ALIVE_VP_CASE(8, 2) ALIVE_VP_CASE(16, 2) ALIVE_VP_CASE(24, 2)
#undef ALIVE_VP_CASE
    ALIVE_CHECK_LAUNCH("alive_debug_valu_pairs");
    return ALIVE_OK;
}

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
    else if (mode == 7) mfma_mixed_rate_kernel<0, 2><<<blocks, 256, 0, s>>>(ri, iters, sink);
    else if (mode == 8) mfma_mixed_rate_kernel<2, 3><<<blocks, 256, 0, s>>>(ri, iters, sink);
    else if (mode == 9) mfma_mixed_rate_kernel<0, 3><<<blocks, 256, 0, s>>>(ri, iters, sink);
    else if (mode == 10) mfma_bf16_feed_kernel<2><<<blocks, 256, 0, s>>>((const unsigned short*)rnd, iters, sink);
    else if (mode == 11) mfma_bf16_feed_kernel<1><<<blocks, 256, 0, s>>>((const unsigned short*)rnd, iters, sink);
    else mfma_rate_kernel<<<blocks, 256, 0, s>>>((const unsigned short*)rnd, iters, mode, sink);
    ALIVE_CHECK_LAUNCH("alive_debug_mfma_rate");
    return ALIVE_OK;
}
