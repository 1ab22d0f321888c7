// The sample-rate and sample-format edges of the path, on the device:
//   torchaudio.functional.resample / gain     inference.py:88-94,135-142 ; realtime_inference.py:146-147,173-175
//   int16 <-> float                           realtime_inference.py:139-140,180-183
// torchaudio is not part of the reference tree; this restates its public algorithm (sinc_interp_hann,
// lowpass_filter_width 6, rolloff 0.99: the polyphase filter bank of _get_sinc_resample_kernel applied as a strided
// convolution) -- "parity unpinned", see DESIGN.md.  HBM-bound streaming kernels, one output sample per thread.
#include "common.h"

namespace {

constexpr int LOWPASS_WIDTH = 6;
constexpr double ROLLOFF = 0.99;

inline int resample_width(int orig, int new_) {
    const double base = (double)(orig < new_ ? orig : new_) * ROLLOFF;
    // ceil(lowpass_filter_width * orig / base)
    double w = (double)LOWPASS_WIDTH * (double)orig / base;
    int wi = (int)w;
    return (double)wi < w ? wi + 1 : wi;
}

// filt[p][j], p < new, j < 2 width + orig: phase p of the windowed sinc, evaluated in fp64 like the host formulation
__global__ void resample_filter_kernel(int orig, int new_, int width, float* __restrict__ filt) {
    const int taps = 2 * width + orig;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= new_ * taps) return;
    const int p = i / taps, j = i - p * taps;
    const double base = (double)(orig < new_ ? orig : new_) * ROLLOFF;
    double t = ((double)(-p) / (double)new_ + (double)(j - width) / (double)orig) * base;
    t = t < -(double)LOWPASS_WIDTH ? -(double)LOWPASS_WIDTH : (t > (double)LOWPASS_WIDTH ? (double)LOWPASS_WIDTH : t);
    const double PI = 3.14159265358979323846;
    const double c = cos(t * PI / (double)LOWPASS_WIDTH / 2.0);
    const double window = c * c;
    t = t * PI;
    const double scale = base / (double)orig;
    const double s = (t == 0.0) ? 1.0 : sin(t) / t;
    filt[i] = (float)(s * window * scale);
}

// y[b][m * new + p] = post * sum_j filt[p][j] * (pre * xpad[b][m * orig + j]),  xpad = x shifted by `width` zeros
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int L, int orig, int new_, int width,
                                                       const float* __restrict__ filt, float pre, float post,
                                                       float* __restrict__ y, int Lout, bool use_lds) {
    extern __shared__ float fs_lds[];             // the whole filter bank when it is small (24k <-> 16k: 46 / 48 floats)
    const int taps = 2 * width + orig;
    const float* fs = filt;                       // large banks (44.1k -> 16k: 160 x 475) stay in global memory / L2
    if (use_lds) {
        for (int i = threadIdx.x; i < new_ * taps; i += blockDim.x) fs_lds[i] = filt[i];
        __syncthreads();
        fs = fs_lds;
    }
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= Lout) return;
    const int b = blockIdx.y;
    const int m = o / new_, p = o - m * new_;
    const float* xb = x + (size_t)b * L;
    const float* f = fs + p * taps;
    const int s0 = m * orig - width;
    float acc = 0.0f;
    for (int j = 0; j < taps; ++j) {
        const int s = s0 + j;
        const float v = (s >= 0 && s < L) ? xb[s] * pre : 0.0f;
        acc = fmaf(f[j], v, acc);
    }
    y[(size_t)b * Lout + o] = acc * post;
}

__global__ void pcm16_to_float_kernel(const short* __restrict__ in, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i] / 32768.0f;
}

// numpy's float32 -> int16 astype on the reference's hosts: truncate toward zero to int32, keep the low 16 bits (no clip)
__global__ void float_to_pcm16_kernel(const float* __restrict__ in, int64_t n, short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (short)(int)(in[i] * 32768.0f);
}

}  // namespace

extern "C" int alive_resample_taps(int orig, int new_) {
    if (orig <= 0 || new_ <= 0) return -1;
    return 2 * resample_width(orig, new_) + orig;
}

extern "C" int64_t alive_resample_length(int64_t L, int orig, int new_) {
    if (orig <= 0 || new_ <= 0 || L < 0) return -1;
    return ((int64_t)new_ * L + orig - 1) / orig;              // ceil(new * L / orig)
}

extern "C" int alive_resample_filter(int orig, int new_, float* filt, void* stream) {
    ALIVE_CHECK_ARG(filt && orig > 0 && new_ > 0, "alive_resample_filter: bad args");
    const int width = resample_width(orig, new_), n = new_ * (2 * width + orig);
    resample_filter_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(orig, new_, width, filt);
    ALIVE_CHECK_LAUNCH("alive_resample_filter");
    return ALIVE_OK;
}

extern "C" int alive_resample(const float* x, int B, int L, int orig, int new_, const float* filt, float pre_scale,
                              float post_scale, float* y, int Lout, void* stream) {
    ALIVE_CHECK_ARG(x && filt && y && B > 0 && L > 0 && orig > 0 && new_ > 0, "alive_resample: bad args");
    ALIVE_CHECK_ARG(Lout > 0 && Lout <= alive_resample_length(L, orig, new_), "alive_resample: Lout %d exceeds ceil(new*L/orig) = %lld",
                    Lout, (long long)alive_resample_length(L, orig, new_));
    const int width = resample_width(orig, new_);
    const size_t bank = (size_t)new_ * (2 * width + orig) * sizeof(float);
    const bool use_lds = bank <= 16 * 1024;
    resample_kernel<<<dim3(cdiv(Lout, 256), B), 256, use_lds ? bank : 0, (hipStream_t)stream>>>(
        x, L, orig, new_, width, filt, pre_scale, post_scale, y, Lout, use_lds);
    ALIVE_CHECK_LAUNCH("alive_resample");
    return ALIVE_OK;
}

extern "C" int alive_pcm16_to_float(const int16_t* in, int64_t n, float* out, void* stream) {
    ALIVE_CHECK_ARG(in && out && n > 0, "alive_pcm16_to_float: bad args");
    pcm16_to_float_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(in, n, out);
    ALIVE_CHECK_LAUNCH("alive_pcm16_to_float");
    return ALIVE_OK;
}

extern "C" int alive_float_to_pcm16(const float* in, int64_t n, int16_t* out, void* stream) {
    ALIVE_CHECK_ARG(in && out && n > 0, "alive_float_to_pcm16: bad args");
    float_to_pcm16_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(in, n, out);
    ALIVE_CHECK_LAUNCH("alive_float_to_pcm16");
    return ALIVE_OK;
}
