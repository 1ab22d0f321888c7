// The waveform-rate edges of the decoder's Filter (/root/reference/module/decoder.py:164,182,186-188,194):
//   source_in  Conv1d(1, 8, 7, pad 3)  followed by  downs[0]  Conv1d(8, 16, 2, stride 2)     -> one kernel
//   source_out Conv1d(8, 1, 7, pad 3)                                                       -> one kernel
// On the implicit-GEMM kernel these ran as K = 7 / 16 / 56 GEMMs padded to MFMA tiles (0.85 ms per 128 windows each);
// they are HBM-bound streaming ops: 1 + 16/2 resp. 8 + 1 floats per sample.  The 8-channel source_in output is not a
// U-Net skip (the skips start at downs[0]), so it never has to exist in memory.  Plain fp32 fmaf chains, accumulator
// starting at the bias, k in (ci, tap) order like the MFMA kernel they replace.
#include "common.h"

namespace {

constexpr int CI = 8, CD = 16, KIN = 7;

// d0[n][co][u] = bd[co] + sum_{ci, j<2} Wd[co][ci][j] * x0[ci][2 u + j],   x0[ci][t] = bin[ci] + sum_m Win[ci][m] * src[t + m - 3]
__global__ __launch_bounds__(256) void source_in_down0_kernel(const float* __restrict__ src, int Lw, const float* __restrict__ Win,
                                                              const float* __restrict__ bin, const float* __restrict__ Wd,
                                                              const float* __restrict__ bd, float* __restrict__ d0) {
    __shared__ float w_in[CI * KIN], b_in[CI], w_d[CD * CI * 2], b_d[CD];
    for (int i = threadIdx.x; i < CI * KIN; i += 256) w_in[i] = Win[i];
    for (int i = threadIdx.x; i < CD * CI * 2; i += 256) w_d[i] = Wd[i];
    if (threadIdx.x < CI) b_in[threadIdx.x] = bin[threadIdx.x];
    if (threadIdx.x < CD) b_d[threadIdx.x] = bd[threadIdx.x];
    __syncthreads();
    const int Lh = Lw / 2;
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= Lh) return;
    const int n = blockIdx.y;
    const float* s = src + (size_t)n * Lw;
    float x[8];                                   // src[2u - 3 .. 2u + 4], zero outside
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int t = 2 * u - 3 + i;
        x[i] = (t >= 0 && t < Lw) ? s[t] : 0.0f;
    }
    float x0[CI][2];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float a = b_in[ci];
#pragma unroll
            for (int m = 0; m < KIN; ++m) a = fmaf(w_in[ci * KIN + m], x[j + m], a);
            x0[ci][j] = a;
        }
    float* o = d0 + (size_t)n * CD * Lh + u;
#pragma unroll
    for (int co = 0; co < CD; ++co) {
        float a = b_d[co];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int j = 0; j < 2; ++j) a = fmaf(w_d[(co * CI + ci) * 2 + j], x0[ci][j], a);
        o[(size_t)co * Lh] = a;
    }
}

// wave[n][t] = b + sum_{ci, m} W[ci][m] * h[ci][t + m - 3]
__global__ __launch_bounds__(256) void source_out_kernel(const float* __restrict__ H, int Lw, const float* __restrict__ W,
                                                         const float* __restrict__ b, float* __restrict__ wave) {
    __shared__ float w[CI * KIN];
    if (threadIdx.x < CI * KIN) w[threadIdx.x] = W[threadIdx.x];
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Lw) return;
    const int n = blockIdx.y;
    const float* h = H + (size_t)n * CI * Lw;
    float a = b[0];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int m = 0; m < KIN; ++m) {
            const int ti = t + m - 3;
            const float v = (ti >= 0 && ti < Lw) ? h[(size_t)ci * Lw + ti] : 0.0f;
            a = fmaf(w[ci * KIN + m], v, a);
        }
    wave[(size_t)n * Lw + t] = a;
}

}  // namespace

extern "C" int alive_filter_source_in(const float* src, int N, int Lw, const float* Win, const float* bin, const float* Wd,
                                      const float* bd, float* d0, void* stream) {
    ALIVE_CHECK_ARG(src && Win && bin && Wd && bd && d0 && N > 0 && Lw >= 2 && (Lw & 1) == 0, "alive_filter_source_in: bad args");
    source_in_down0_kernel<<<dim3(cdiv(Lw / 2, 256), N), 256, 0, (hipStream_t)stream>>>(src, Lw, Win, bin, Wd, bd, d0);
    ALIVE_CHECK_LAUNCH("alive_filter_source_in");
    return ALIVE_OK;
}

extern "C" int alive_filter_source_out(const float* H, int N, int Lw, const float* W, const float* b, float* wave, void* stream) {
    ALIVE_CHECK_ARG(H && W && b && wave && N > 0 && Lw > 0, "alive_filter_source_out: bad args");
    source_out_kernel<<<dim3(cdiv(Lw, 256), N), 256, 0, (hipStream_t)stream>>>(H, Lw, W, b, wave);
    ALIVE_CHECK_LAUNCH("alive_filter_source_out");
    return ALIVE_OK;
}
