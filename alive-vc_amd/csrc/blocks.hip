// Channel-wise blocks of the ConvNeXt stacks and the small glue kernels.
//   dwconv7 + ChannelNorm / AdaptiveChannelNorm   module/common.py:20-26,35-41,55-56,75-76
//   argmax over classes                           module/f0_estimator.py:33
//   pitch transform                               inference.py:119-126,130 / realtime_inference.py:156-163
//   |re,im| -> magnitude                          module/spectrogram.py:8
// HBM-bound kernels: lanes run along time (coalesced 256-B rows), channels are
// split over the 4 waves of a block and reduced through LDS.
#include "common.h"

namespace {

constexpr int TT = 64;   // time columns per block (one per lane)

// Y = affine( (dw(X) - mean_C) / (std_C(unbiased) + eps) );  DW=false: plain ChannelNorm
template <bool DW>
__global__ __launch_bounds__(256) void dwconv_norm_kernel(
    const float* __restrict__ X, int C, int T, const float* __restrict__ dw_w, const float* __restrict__ dw_b,
    int affine_mode, const float* __restrict__ gain, const float* __restrict__ offset,
    const float* __restrict__ cond, int cond_rows, int scale_row, int shift_row, float eps, float* __restrict__ Y) {
    __shared__ float red[4][TT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const int t = blockIdx.x * TT + lane;
    const bool ok = t < T;
    const float* Xn = X + (size_t)n * C * T;
    float* Yn = Y + (size_t)n * C * T;

    // pass 1: depthwise conv (zero pad 3), running sum
    float s = 0.0f;
    for (int c = wv; c < C; c += 4) {
        float y = 0.0f;
        if (ok) {
            const float* xc = Xn + (size_t)c * T;
            if (DW) {
                y = dw_b[c];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    int ti = t + j - 3;
                    float xv = (ti >= 0 && ti < T) ? xc[ti] : 0.0f;
                    y = fmaf(dw_w[c * 7 + j], xv, y);
                }
                Yn[(size_t)c * T + t] = y;
            } else {
                y = xc[t];
            }
        }
        s += y;
    }
    red[wv][lane] = s;
    __syncthreads();
    const float mean = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)C;
    __syncthreads();
    // pass 2: centred sum of squares (each thread re-reads only what it wrote)
    const float* src = DW ? Yn : Xn;
    float ss = 0.0f;
    for (int c = wv; c < C; c += 4) {
        if (ok) {
            float d = src[(size_t)c * T + t] - mean;
            ss = fmaf(d, d, ss);
        }
    }
    red[wv][lane] = ss;
    __syncthreads();
    const float var = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)(C - 1);
    const float sigma = sqrtf(var) + eps;
    // pass 3: normalise + affine
    if (!ok) return;
    for (int c = wv; c < C; c += 4) {
        float v = (src[(size_t)c * T + t] - mean) / sigma;
        float g, o;
        if (affine_mode == 0) {
            g = gain[c];
            o = offset[c];
        } else {
            g = cond[((size_t)n * cond_rows + scale_row + c) * T + t];
            o = cond[((size_t)n * cond_rows + shift_row + c) * T + t];
        }
        Yn[(size_t)c * T + t] = v * g + o;
    }
}

// ---- small-T variants (streaming: a handful of frames): one block per (frame, window), threads along the channels ----
__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

template <bool DW>
__global__ __launch_bounds__(256) void dwconv_norm_small_kernel(
    const float* __restrict__ X, int C, int T, const float* __restrict__ dw_w, const float* __restrict__ dw_b,
    int affine_mode, const float* __restrict__ gain, const float* __restrict__ offset,
    const float* __restrict__ cond, int cond_rows, int scale_row, int shift_row, float eps, float* __restrict__ Y) {
    __shared__ float sh[4];
    const int t = blockIdx.x, n = blockIdx.y;
    const float* Xn = X + (size_t)n * C * T;
    float* Yn = Y + (size_t)n * C * T;
    float s = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float* xc = Xn + (size_t)c * T;
        float y;
        if (DW) {
            y = dw_b[c];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                int ti = t + j - 3;
                float xv = (ti >= 0 && ti < T) ? xc[ti] : 0.0f;
                y = fmaf(dw_w[c * 7 + j], xv, y);
            }
            Yn[(size_t)c * T + t] = y;
        } else {
            y = xc[t];
        }
        s += y;
    }
    const float mean = block_sum(s, sh) / (float)C;
    const float* src = DW ? Yn : Xn;
    float ss = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float d = src[(size_t)c * T + t] - mean;
        ss = fmaf(d, d, ss);
    }
    const float sigma = sqrtf(block_sum(ss, sh) / (float)(C - 1)) + eps;
    for (int c = threadIdx.x; c < C; c += 256) {
        float v = (src[(size_t)c * T + t] - mean) / sigma;
        float g, o;
        if (affine_mode == 0) { g = gain[c]; o = offset[c]; }
        else {
            g = cond[((size_t)n * cond_rows + scale_row + c) * T + t];
            o = cond[((size_t)n * cond_rows + shift_row + c) * T + t];
        }
        Yn[(size_t)c * T + t] = v * g + o;
    }
}

__global__ __launch_bounds__(256) void argmax_small_kernel(const float* __restrict__ X, int C, int T, float* __restrict__ out) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const int t = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const float* Xn = X + (size_t)n * C * T + t;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int c = tid; c < C; c += 256) {
        float v = Xn[(size_t)c * T];
        if (v > best || (v != v && best == best)) { best = v; besti = c; }
    }
    bv[tid] = best;
    bi[tid] = besti;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            float v = bv[tid + o];
            int vi = bi[tid + o];
            bool take = (v > bv[tid]) || (v == bv[tid] && vi < bi[tid]) || (v != v && bv[tid] == bv[tid]);
            if (take) { bv[tid] = v; bi[tid] = vi; }
        }
        __syncthreads();
    }
    if (tid == 0) out[(size_t)n * T + t] = (float)bi[0];
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ X, int C, int T, float* __restrict__ out) {
    __shared__ float bv[4][TT];
    __shared__ int bi[4][TT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const int t = blockIdx.x * TT + lane;
    const bool ok = t < T;
    const float* Xn = X + (size_t)n * C * T;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    if (ok) {
        for (int c = wv; c < C; c += 4) {
            float v = Xn[(size_t)c * T + t];
            if (v > best || (v != v && best == best)) { best = v; besti = c; }   // first max; NaN wins like ATen
        }
    }
    bv[wv][lane] = best;
    bi[wv][lane] = besti;
    __syncthreads();
    if (wv == 0 && ok) {
        float b = bv[0][lane];
        int i = bi[0][lane];
        for (int w = 1; w < 4; ++w) {
            float v = bv[w][lane];
            int vi = bi[w][lane];
            bool take = (v > b) || (v == b && vi < i) || (v != v && b == b);
            if (take) { b = v; i = vi; }
        }
        out[(size_t)n * T + t] = (float)i;
    }
}

// One block per window.  pitch = 12*log2(f0/440) - 9, mean over finite entries, intonation / shift,
// back to Hz, NaN/inf -> 0, * f0_rate.  log2/exp2 are evaluated in fp64 and rounded once, the closest
// a device kernel can get to the 1-ulp vector libm of the CPU path.
__global__ __launch_bounds__(256) void pitch_kernel(float* f0, int T, int mode, float f0_rate, float shift, float inton) {
    __shared__ double s_sum[256];
    __shared__ int s_cnt[256];
    float* f = f0 + (size_t)blockIdx.x * T;
    const int tid = threadIdx.x;
    double sum = 0.0;
    int cnt = 0;
    if (mode == 0) {
        for (int t = tid; t < T; t += 256) {
            float p = 12.0f * (float)log2((double)(f[t] / 440.0f)) - 9.0f;
            if (!isinf(p) && !isnan(p)) { sum += (double)p; ++cnt; }
        }
        s_sum[tid] = sum;
        s_cnt[tid] = cnt;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) { s_sum[tid] += s_sum[tid + o]; s_cnt[tid] += s_cnt[tid + o]; }
            __syncthreads();
        }
    }
    const float mean = (mode == 0) ? (float)(s_sum[0] / (double)s_cnt[0]) : 0.0f;
    for (int t = tid; t < T; t += 256) {
        float x = f[t];
        if (mode == 1) x = x * f0_rate;
        float p = 12.0f * (float)log2((double)(x / 440.0f)) - 9.0f;
        if (mode == 0) p = mean + (p - mean) * inton + shift;
        else p = p + shift;
        float e = (p + 9.0f) / 12.0f;
        float y = 440.0f * (float)exp2((double)e);
        if (isnan(y) || isinf(y)) y = 0.0f;
        if (mode == 0) y = y * f0_rate;
        f[t] = y;
    }
}

// re/im planes [N][2*B][T] (rows 0..B-1 real, B..2B-1 imag) -> magnitude [N][B][T]
__global__ void magnitude_kernel(const float* __restrict__ ri, int B, int T, float* __restrict__ out, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int t = (int)(i % T);
    size_t r = i / T;
    int b = (int)(r % B);
    size_t n = r / B;
    float re = ri[((size_t)n * 2 * B + b) * T + t];
    float im = ri[((size_t)n * 2 * B + B + b) * T + t];
    out[i] = hypotf(re, im);
}

}  // namespace

extern "C" int alive_dwconv_norm(const float* X, int N, int C, int T, const float* dw_w, const float* dw_b,
                                 int affine_mode, const float* gain, const float* offset, const float* cond,
                                 int cond_rows, int scale_row, int shift_row, float eps, float* Y, void* stream) {
    ALIVE_CHECK_ARG(X && Y && dw_w && dw_b && N > 0 && C > 1 && T > 0, "alive_dwconv_norm: bad args");
    ALIVE_CHECK_ARG(X != Y, "alive_dwconv_norm: in-place not supported");
    ALIVE_CHECK_ARG(affine_mode == 0 ? (gain && offset) : (cond != nullptr), "alive_dwconv_norm: affine params");
    if (T <= 32) {      // streaming: one block per frame, threads along the channels
        dwconv_norm_small_kernel<true><<<dim3(T, N), 256, 0, (hipStream_t)stream>>>(X, C, T, dw_w, dw_b, affine_mode, gain, offset,
                                                                                   cond, cond_rows, scale_row, shift_row, eps, Y);
        ALIVE_CHECK_LAUNCH("alive_dwconv_norm");
        return ALIVE_OK;
    }
    dim3 g(cdiv(T, TT), N);
    dwconv_norm_kernel<true><<<g, 256, 0, (hipStream_t)stream>>>(X, C, T, dw_w, dw_b, affine_mode, gain, offset, cond,
                                                                cond_rows, scale_row, shift_row, eps, Y);
    ALIVE_CHECK_LAUNCH("alive_dwconv_norm");
    return ALIVE_OK;
}

extern "C" int alive_channel_norm(const float* X, int N, int C, int T, const float* gain, const float* offset, float eps,
                                  float* Y, void* stream) {
    ALIVE_CHECK_ARG(X && Y && gain && offset && N > 0 && C > 1 && T > 0, "alive_channel_norm: bad args");
    if (T <= 32) {
        dwconv_norm_small_kernel<false><<<dim3(T, N), 256, 0, (hipStream_t)stream>>>(X, C, T, nullptr, nullptr, 0, gain, offset,
                                                                                    nullptr, 0, 0, 0, eps, Y);
        ALIVE_CHECK_LAUNCH("alive_channel_norm");
        return ALIVE_OK;
    }
    dim3 g(cdiv(T, TT), N);
    dwconv_norm_kernel<false><<<g, 256, 0, (hipStream_t)stream>>>(X, C, T, nullptr, nullptr, 0, gain, offset, nullptr, 0, 0,
                                                                 0, eps, Y);
    ALIVE_CHECK_LAUNCH("alive_channel_norm");
    return ALIVE_OK;
}

extern "C" int alive_argmax_channels(const float* X, int N, int C, int T, float* out, void* stream) {
    ALIVE_CHECK_ARG(X && out && N > 0 && C > 0 && T > 0, "alive_argmax_channels: bad args");
    if (T <= 32) {
        argmax_small_kernel<<<dim3(T, N), 256, 0, (hipStream_t)stream>>>(X, C, T, out);
        ALIVE_CHECK_LAUNCH("alive_argmax_channels");
        return ALIVE_OK;
    }
    dim3 g(cdiv(T, TT), N);
    argmax_kernel<<<g, 256, 0, (hipStream_t)stream>>>(X, C, T, out);
    ALIVE_CHECK_LAUNCH("alive_argmax_channels");
    return ALIVE_OK;
}

extern "C" int alive_pitch_transform(float* f0, int N, int T, int mode, float f0_rate, float pitch_shift, float intonation,
                                     void* stream) {
    ALIVE_CHECK_ARG(f0 && N > 0 && T > 0 && (mode == 0 || mode == 1), "alive_pitch_transform: bad args");
    pitch_kernel<<<N, 256, 0, (hipStream_t)stream>>>(f0, T, mode, f0_rate, pitch_shift, intonation);
    ALIVE_CHECK_LAUNCH("alive_pitch_transform");
    return ALIVE_OK;
}

int alive_magnitude(const float* ri, int N, int B, int T, float* out, hipStream_t s) {
    size_t total = (size_t)N * B * T;
    magnitude_kernel<<<cdiv(total, 256), 256, 0, s>>>(ri, B, T, out, total);
    ALIVE_CHECK_LAUNCH("alive_magnitude");
    return ALIVE_OK;
}
