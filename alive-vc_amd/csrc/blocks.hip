// Channel-wise blocks of the ConvNeXt stacks and the small glue kernels.
//   dwconv7 + ChannelNorm / AdaptiveChannelNorm   module/common.py:20-26,35-41,55-56,75-76
//   argmax over classes                           module/f0_estimator.py:33
//   pitch transform                               inference.py:119-126,130 / realtime_inference.py:156-163
//   |re,im| -> magnitude                          module/spectrogram.py:8
// HBM-bound kernels: lanes run along time (coalesced 256-B rows), channels are
// split over the 4 waves of a block and reduced through LDS.
#include "conv_epilogue.h"
#include "planes_layout.h"

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

// ---- the same, single pass, plane-packed output for the GEMM that follows (gemm_planes.hip) ----
// The kernel above writes dw(x) to memory and re-reads it twice for the statistics and the affine (three passes at
// ~1.1 TB/s), and alive_to_planes then re-reads the result to split it.  Here a block keeps its [C][64 columns] tile of
// dw(x) in LDS (133 KB at C = 512), takes mean / sigma from it, applies the affine in place (lanes along time, so the
// per-sample scale / shift rows of the adaptive form are read coalesced) and leaves through an LDS transpose as bf16
// planes (k-blocked, planes_layout.h): HBM sees x once and the planes once.
constexpr int NPT = 512;                 // 8 waves: channels are split over the waves
// TW = columns per tile: 64 (a wave row is one channel) or 32 (a wave row is two channels; halves the LDS tile so that two
// blocks share a CU at C = 512 -- with one block of 8 waves per CU the passes are latency-bound)
template <int NP, int TW>
__global__ __launch_bounds__(NPT) void dwconv_norm_planes_kernel(
    const float* __restrict__ X, int C, int T, const float* __restrict__ dw_w, const float* __restrict__ dw_b, int dw,
    int affine_mode, const float* __restrict__ gain, const float* __restrict__ offset, const float* __restrict__ cond,
    int cond_rows, int scale_row, int shift_row, float eps, unsigned short* __restrict__ P, int64_t cols_pad, float f16_scale) {
    // f16_scale != 0 (NP = 2): the planes are fp16 (hi, lo) of f16_scale * value (AliveGemm.f16s), else bf16 (NP = 1: one fp16 plane)
    extern __shared__ float tile[];                      // [C][TW + 1]
    constexpr int CPW = 64 / TW;                         // channels per wave row
    constexpr int NR = NPT / 64 * CPW;                   // channel rows in flight per block
    __shared__ float red[NR][TW];
    constexpr int PT = TW + 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tl = lane % TW, crow = wv * CPW + lane / TW;
    const int n = blockIdx.y, t0 = blockIdx.x * TW;
    const int t = t0 + tl;
    const bool ok = t < T;
    const float* Xn = X + (size_t)n * C * T;
    // pass 1: depthwise conv (zero pad 3) -> LDS, running sum
    float s = 0.0f;
    if (dw && t0 >= 3 && t0 + TW + 3 <= T) {
        // interior tile (block-uniform): no tap leaves the window -- the same loads and fmas without the seven bounds tests per element
        // (the kernel is bound by its instruction count, not by memory: tools/experiments/README.md)
        const float* xt = Xn + t - 3;
        for (int c = crow; c < C; c += NR) {
            const float* xc = xt + (size_t)c * T;
            const float* wc = dw_w + c * 7;
            float y = dw_b[c];
#pragma unroll
            for (int j = 0; j < 7; ++j) y = fmaf(wc[j], xc[j], y);
            tile[c * PT + tl] = y;
            s += y;
        }
    } else
    for (int c = crow; c < C; c += NR) {
        float y = 0.0f;
        if (ok) {
            const float* xc = Xn + (size_t)c * T;
            if (dw) {
                y = dw_b[c];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int ti = t + j - 3;
                    const float xv = (ti >= 0 && ti < T) ? xc[ti] : 0.0f;
                    y = fmaf(dw_w[c * 7 + j], xv, y);
                }
            } else {
                y = xc[t];
            }
        }
        tile[c * PT + tl] = y;
        s += y;
    }
    red[crow][tl] = s;
    __syncthreads();
    float mean = 0.0f;
#pragma unroll
    for (int i = 0; i < NR; ++i) mean += red[i][tl];
    mean = mean / (float)C;
    __syncthreads();
    // pass 2: centred sum of squares
    float ss = 0.0f;
    for (int c = crow; c < C; c += NR) {
        const float d = tile[c * PT + tl] - mean;
        ss = fmaf(d, d, ss);
    }
    red[crow][tl] = ok ? ss : 0.0f;
    __syncthreads();
    float var = 0.0f;
#pragma unroll
    for (int i = 0; i < NR; ++i) var += red[i][tl];
    const float sigma = sqrtf(var / (float)(C - 1)) + eps;
    // pass 3: normalise + affine, in place
    for (int c = crow; c < C; c += NR) {
        const float v = (tile[c * PT + tl] - mean) / sigma;
        float g = 1.0f, o = 0.0f;
        if (affine_mode == 0) {
            g = gain[c];
            o = offset[c];
        } else if (ok) {
            g = cond[((size_t)n * cond_rows + scale_row + c) * T + t];
            o = cond[((size_t)n * cond_rows + shift_row + c) * T + t];
        }
        tile[c * PT + tl] = ok ? v * g + o : 0.0f;
    }
    __syncthreads();
    // pass 4: [C][col] -> k-blocked planes (planes_layout.h): thread = (k-block, column, 8-channel chunk of the block); 4 TW
    // consecutive threads write TW x 64 B = one contiguous run
    const int chunks = C / 8;
    for (int item = threadIdx.x; item < TW * chunks; item += NPT) {
        const int cl = (item >> 2) % TW, ck = (item / (4 * TW)) * 4 + (item & 3);
        if (t0 + cl >= T) continue;
        float vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = tile[(ck * 8 + e) * PT + cl];
        const bool f16s = NP == 2 && f16_scale != 0.0f;
        if (f16s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[e] *= f16_scale;
        }
        const int64_t col = (int64_t)n * T + t0 + cl;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            u32x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                const bf16x2_t hp = {(__bf16)vv[2 * e], (__bf16)vv[2 * e + 1]};
                const unsigned h = (NP == 1 || f16s) ? pack_f16x2(vv[2 * e], vv[2 * e + 1]) : __builtin_bit_cast(unsigned, hp);     // NP = 1: one fp16 plane
                o4[e] = h;
                if (f16s) {
                    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                    const f16x2_t hv = __builtin_bit_cast(f16x2_t, h);
                    vv[2 * e] -= (float)hv[0];
                    vv[2 * e + 1] -= (float)hv[1];
                } else {
                    vv[2 * e] -= __uint_as_float(h << 16);
                    vv[2 * e + 1] -= __uint_as_float(h & 0xffff0000u);
                }
            }
            *(u32x4*)(P + planes_at(pl, col, ck * 8, cols_pad, C)) = o4;
        }
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

// dw conv (or none: dw_w == NULL) + (Adaptive)ChannelNorm with plane-packed output; C a multiple of 32
namespace {
int dwconv_norm_planes_impl(const float* X, int N, int C, int T, const float* dw_w, const float* dw_b,
                            int affine_mode, const float* gain, const float* offset, const float* cond,
                            int cond_rows, int scale_row, int shift_row, float eps, int planes, float f16_scale, void* P,
                            void* stream) {
    ALIVE_CHECK_ARG(X && P && N > 0 && C > 1 && T > 0, "alive_dwconv_norm_planes: bad args");
    ALIVE_CHECK_ARG((dw_w == nullptr) == (dw_b == nullptr), "alive_dwconv_norm_planes: dw_w and dw_b go together");
    ALIVE_CHECK_ARG(affine_mode == 0 ? (gain && offset) : (cond != nullptr), "alive_dwconv_norm_planes: affine params");
    ALIVE_CHECK_ARG((C & 31) == 0 && planes >= 1 && planes <= 3, "alive_dwconv_norm_planes: C %d must be a multiple of 32, planes 1, 2 or 3", C);
    const int tw = C > 256 ? 32 : 64;                    // <= 68 KB of LDS per block: two blocks per CU
    const size_t lds = (size_t)C * (tw + 1) * sizeof(float);
    ALIVE_CHECK_ARG(lds <= 150 * 1024, "alive_dwconv_norm_planes: C %d does not fit the LDS tile", C);
    const int64_t cols = (int64_t)N * T, cols_pad = (cols + 127) / 128 * 128;
    {
        static LdsOptIn optin;
        hipError_t e = optin.ensure({(const void*)dwconv_norm_planes_kernel<2, 32>, (const void*)dwconv_norm_planes_kernel<3, 32>,
                                     (const void*)dwconv_norm_planes_kernel<2, 64>, (const void*)dwconv_norm_planes_kernel<3, 64>,
                                     (const void*)dwconv_norm_planes_kernel<1, 32>, (const void*)dwconv_norm_planes_kernel<1, 64>}, 150 * 1024);
        if (e != hipSuccess) {
            alive_set_error("alive_dwconv_norm_planes: cannot reserve LDS: %s", hipGetErrorString(e));
            return ALIVE_ERR_LAUNCH;
        }
    }
    dim3 g(cdiv(T, tw), N);
#define LAUNCH_DNP(NP_, TW_)                                                                                                   \
    dwconv_norm_planes_kernel<NP_, TW_><<<g, NPT, lds, (hipStream_t)stream>>>(X, C, T, dw_w, dw_b, dw_w != nullptr, affine_mode, gain, \
                                                                             offset, cond, cond_rows, scale_row, shift_row, eps,  \
                                                                             (unsigned short*)P, cols_pad, f16_scale)
    if (planes == 1 && tw == 32) LAUNCH_DNP(1, 32);         // one fp16 plane: the input of a plain GEMM (alive_gemm_planes, planes = 1)
    else if (planes == 1) LAUNCH_DNP(1, 64);
    else if (planes == 2 && tw == 32) LAUNCH_DNP(2, 32);
    else if (planes == 3 && tw == 32) LAUNCH_DNP(3, 32);
    else if (planes == 2) LAUNCH_DNP(2, 64);
    else LAUNCH_DNP(3, 64);
#undef LAUNCH_DNP
    ALIVE_CHECK_LAUNCH("alive_dwconv_norm_planes");
    return ALIVE_OK;
}
}  // namespace

extern "C" int alive_dwconv_norm_planes(const float* X, int N, int C, int T, const float* dw_w, const float* dw_b,
                                        int affine_mode, const float* gain, const float* offset, const float* cond,
                                        int cond_rows, int scale_row, int shift_row, float eps, int planes, void* P,
                                        void* stream) {
    return dwconv_norm_planes_impl(X, N, C, T, dw_w, dw_b, affine_mode, gain, offset, cond, cond_rows, scale_row, shift_row, eps, planes, 0.0f,
                                   P, stream);
}

extern "C" int alive_dwconv_norm_planes_f16s(const float* X, int N, int C, int T, const float* dw_w, const float* dw_b,
                                             int affine_mode, const float* gain, const float* offset, const float* cond,
                                             int cond_rows, int scale_row, int shift_row, float eps, float scale, void* P,
                                             void* stream) {
    ALIVE_CHECK_ARG(scale > 0.0f, "alive_dwconv_norm_planes_f16s: scale must be a positive power of two");
    return dwconv_norm_planes_impl(X, N, C, T, dw_w, dw_b, affine_mode, gain, offset, cond, cond_rows, scale_row, shift_row, eps, 2, scale,
                                   P, stream);
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

// ---- z = gelu(h) * interp(scale) + interp(shift): the input of a FilterResBlock's first modulated conv (decoder.py:112-117,130-132) ----
// Since round 4 the Filter's coarsest up conv and the 1x1 input conv of its FilterBlock are ONE transposed conv (their weights composed
// at checkpoint load: both are linear and nothing sits between them, decoder.py:147,192-193), so the modulated copy of its output, which
// the input conv's epilogue used to emit, is made here -- with that epilogue's arithmetic (conv_split.hip: gelu_fast2, ATen's lerp,
// multiply then add) and in its formats: fp32 [N][C][L] or k-blocked planes for the split kernel's plane input.
namespace {
template <bool PLANES>
__global__ __launch_bounds__(256) void gelu_film_kernel(const float* __restrict__ H, int C, int L, const float* __restrict__ film,
                                                        int film_rows, int Lf, int scale_row, int shift_row, int t_off, int f_off,
                                                        int film_ld, float ratio, float* __restrict__ Z,
                                                        unsigned short* __restrict__ Zp, int64_t cols_pad, int n_planes) {
    __shared__ float tile[PLANES ? 64 : 1][PLANES ? 65 : 1];       // [channel][column]
    const int tid = threadIdx.x;
    const int n = blockIdx.z, c0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    const int t4 = t0 + (tid & 15) * 4;                             // this thread's four columns
    Lerp lp[4];
    int fc0[4], fc1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = t4 + q < L ? t4 + q : L - 1;
        lp[q] = lerp_coord(t + t_off, ratio, Lf);
        int a = lp[q].i0 - f_off, b = lp[q].i1 - f_off;             // frame of the window -> column of the film tensor
        fc0[q] = a < 0 ? 0 : (a < film_ld ? a : film_ld - 1);
        fc1[q] = b < 0 ? 0 : (b < film_ld ? b : film_ld - 1);
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int cl = pass * 16 + (tid >> 4), c = c0 + cl;
        if (c >= C) continue;
        const size_t o = ((size_t)n * C + c) * L + t4;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        const bool vec = (L & 3) == 0 && t4 + 3 < L;               // rows are 16-byte aligned only when L is a multiple of 4
        if (vec) v = *(const f32x4*)(H + o);
        else for (int q = 0; q < 4 && t4 + q < L; ++q) v[q] = H[o + q];
        const f32x2 g0 = gelu_fast2(f32x2{v[0], v[1]}), g1 = gelu_fast2(f32x2{v[2], v[3]});
        const float gv[4] = {g0[0], g0[1], g1[0], g1[1]};
        const float* fs = film + ((size_t)n * film_rows + scale_row + c) * film_ld;
        const float* fh = film + ((size_t)n * film_rows + shift_row + c) * film_ld;
        f32x4 z;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float sc = lerp_apply(lp[q], fs[fc0[q]], fs[fc1[q]]);
            const float sh = lerp_apply(lp[q], fh[fc0[q]], fh[fc1[q]]);
            z[q] = gv[q] * sc + sh;
        }
        if constexpr (PLANES) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tile[cl][(tid & 15) * 4 + q] = z[q];
        } else {
            if (vec) *(f32x4*)(Z + o) = z;
            else for (int q = 0; q < 4 && t4 + q < L; ++q) Z[o + q] = z[q];
        }
    }
    if constexpr (PLANES) {
        __syncthreads();
        // thread -> (k-block half of the 64 channels, column, 8-channel chunk): a wave writes 16 columns x 64 B = 1 KB per plane
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int ck = it * 4 + (tid & 3), cl = (tid >> 2) & 63;
            const int t = t0 + cl, c = c0 + ck * 8;
            if (t >= L || c >= C) continue;
            const int64_t col = (int64_t)n * L + t;
            float vv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[e] = tile[ck * 8 + e][cl];
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                if (pl >= n_planes) break;                         // one plane (fp16): the plain consumer (AliveConv.precision 3)
                u32x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16x2_t hp = {(__bf16)vv[2 * e], (__bf16)vv[2 * e + 1]};
                    const unsigned h = n_planes == 1 ? pack_f16x2(vv[2 * e], vv[2 * e + 1]) : __builtin_bit_cast(unsigned, hp);
                    o4[e] = h;
                    vv[2 * e] -= __uint_as_float(h << 16);
                    vv[2 * e + 1] -= __uint_as_float(h & 0xffff0000u);
                }
                *(u32x4*)(Zp + planes_at(pl, col, c, cols_pad, C)) = o4;
            }
        }
    }
}
}  // namespace

// planes: 2 (split bf16), or 1 = ONE fp16 plane (the input of a plain conv, AliveConv.precision 3)
int alive_gelu_film_impl(const float* H, int N, int C, int L, const float* film, int film_rows, int Lf, int scale_row,
                         int shift_row, int t0, int f0, int film_ld, float* Z, void* Zp, int planes, void* stream) {
    ALIVE_CHECK_ARG(H && film && (Z != nullptr) != (Zp != nullptr) && N > 0 && C > 0 && L > 0 && Lf > 0, "alive_gelu_film: bad args (one of Z / Zp)");
    ALIVE_CHECK_ARG(film_ld > 0 && t0 >= 0 && f0 >= 0, "alive_gelu_film: bad frame range");
    ALIVE_CHECK_ARG(((((uintptr_t)H) | ((uintptr_t)Z) | ((uintptr_t)Zp)) & 15) == 0, "alive_gelu_film: H / Z / Zp must be 16-byte aligned");
    ALIVE_CHECK_ARG(Zp == nullptr || (C & 31) == 0, "alive_gelu_film: plane output needs C %% 32 == 0");
    const float ratio = (float)film_ld / (float)L;           // == window frames / window samples at this rate
    dim3 g(cdiv(L, 64), cdiv(C, 64), N);
    const int64_t cols_pad = ((int64_t)N * L + 127) / 128 * 128;
    if (Zp) gelu_film_kernel<true><<<g, 256, 0, (hipStream_t)stream>>>(H, C, L, film, film_rows, Lf, scale_row, shift_row, t0, f0, film_ld, ratio,
                                                                     nullptr, (unsigned short*)Zp, cols_pad, planes);
    else gelu_film_kernel<false><<<g, 256, 0, (hipStream_t)stream>>>(H, C, L, film, film_rows, Lf, scale_row, shift_row, t0, f0, film_ld, ratio, Z,
                                                                    nullptr, 0, 0);
    ALIVE_CHECK_LAUNCH("alive_gelu_film");
    return ALIVE_OK;
}

extern "C" int alive_gelu_film(const float* H, int N, int C, int L, const float* film, int film_rows, int Lf, int scale_row,
                               int shift_row, int t0, int f0, int film_ld, float* Z, void* Zp, void* stream) {
    return alive_gelu_film_impl(H, N, C, L, film, film_rows, Lf, scale_row, shift_row, t0, f0, film_ld, Z, Zp, 2, stream);
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

ALIVE_F16_SAT_GETTER(alive_f16_sat_blocks)
