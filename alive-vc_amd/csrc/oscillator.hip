// HarmonicOscillator.forward after to_amps/exp -- /root/reference/module/decoder.py:79-100.
//
//   formants[h][f] = f0[f] * (h+1)                       (fp32 product)
//   F[h][u]  = linear-interp x320 of formants            (fma(w0, a, round(w1*b)), fmaf coordinate)
//   dt[h][u] = fp32( sum_{v<=u} double(F[h][v] / 16000) )   <- ATen's CPU cumsum: fp64 accumulate,
//                                                             every prefix rounded to fp32
//   dt -= dt[crop0];  theta = 2pi*dt + phi;  wave[u] = mean_h( sin(theta) * A[h][u] )
//
// The phase reaches ~2e5 cycles inside a 9 s window, where one fp32 ulp is 1/64
// cycle, so the kernel reproduces the reference's arithmetic step by step
// instead of using a better-conditioned phase accumulator.
//
// Three launches: per-frame fp64 segment sums -> per-(n,h) exclusive scan over
// frames (+ the fp32 phase at crop0) -> synthesis.  In the synthesis kernel a
// wave owns one frame: lane = harmonic walks the frame's 320 samples with the
// fp64 running sum in a register pair, 64-sample batches are transposed
// through LDS so that lane = sample sums the harmonics in index order.
#include "common.h"

namespace {

struct OscGeom {
    int H, Lf, seg, Lw;
    float ratio;        // (float)Lf / Lw
    float sample_rate;
};

__device__ __forceinline__ float formant_step(const float* f0n, float hmul, const Lerp& l, float sample_rate) {
    float a = f0n[l.i0] * hmul;
    float b = f0n[l.i1] * hmul;
    return __fdiv_rn(lerp_apply(l, a, b), sample_rate);
}

__global__ __launch_bounds__(64) void osc_segsum_kernel(const float* __restrict__ f0, OscGeom g, double* __restrict__ S) {
    const int f = blockIdx.x, n = blockIdx.y;
    const int h = blockIdx.z * 64 + threadIdx.x;
    if (h >= g.H) return;
    const float* f0n = f0 + (size_t)n * g.Lf;
    const float hmul = (float)(h + 1);
    double acc = 0.0;
    const int u0 = f * g.seg;
    for (int i = 0; i < g.seg; ++i) {
        Lerp l = lerp_coord(u0 + i, g.ratio, g.Lf);
        acc += (double)formant_step(f0n, hmul, l, g.sample_rate);
    }
    S[((size_t)n * g.H + h) * g.Lf + f] = acc;
}

// exclusive scan of the segment sums (in place) and dt0 = fp32 prefix at crop0
__global__ __launch_bounds__(64) void osc_prefix_kernel(const float* __restrict__ f0, OscGeom g, int crop0,
                                                        double* __restrict__ S, float* __restrict__ dt0) {
    const int n = blockIdx.x;
    const int h = blockIdx.y * 64 + threadIdx.x;
    if (h >= g.H) return;
    double* s = S + ((size_t)n * g.H + h) * g.Lf;
    double run = 0.0;
    const int fc = crop0 / g.seg;
    double at_fc = 0.0;
    for (int f = 0; f < g.Lf; ++f) {
        double v = s[f];
        s[f] = run;
        if (f == fc) at_fc = run;
        run += v;
    }
    const float* f0n = f0 + (size_t)n * g.Lf;
    const float hmul = (float)(h + 1);
    double acc = at_fc;
    for (int u = fc * g.seg; u <= crop0; ++u) {
        Lerp l = lerp_coord(u, g.ratio, g.Lf);
        acc += (double)formant_step(f0n, hmul, l, g.sample_rate);
    }
    dt0[(size_t)n * g.H + h] = (float)acc;
}

__global__ __launch_bounds__(64) void osc_synth_kernel(const float* __restrict__ amps, const float* __restrict__ f0,
                                                       const float* __restrict__ phi_in, OscGeom g,
                                                       const double* __restrict__ P, const float* __restrict__ dt0,
                                                       int phi_col, float* __restrict__ wave, float* __restrict__ phi_out) {
    __shared__ float tile[64][65];
    const int f = blockIdx.x, n = blockIdx.y;
    const int lane = threadIdx.x;
    const int h = lane;
    const bool hv = h < g.H;
    const float* f0n = f0 + (size_t)n * g.Lf;
    const float* an = amps + ((size_t)n * g.H + (hv ? h : 0)) * g.Lf;
    const float hmul = (float)(h + 1);
    const float TWO_PI_F = 6.283185307179586f;
    double acc = hv ? P[((size_t)n * g.H + h) * g.Lf + f] : 0.0;
    const float d0 = hv ? dt0[(size_t)n * g.H + h] : 0.0f;
    const float ph = (hv && phi_in != nullptr) ? phi_in[(size_t)n * g.H + h] : 0.0f;
    const int u0 = f * g.seg;
    for (int b0 = 0; b0 < g.seg; b0 += 64) {
        const int nb = (g.seg - b0) < 64 ? (g.seg - b0) : 64;
        for (int i = 0; i < nb; ++i) {
            const int u = u0 + b0 + i;
            Lerp l = lerp_coord(u, g.ratio, g.Lf);
            float c = 0.0f;
            if (hv) {
                acc += (double)formant_step(f0n, hmul, l, g.sample_rate);
                float dt = (float)acc - d0;
                float theta = __fadd_rn(__fmul_rn(TWO_PI_F, dt), ph);
                float sn = sinf(theta);
                float a = lerp_apply(l, an[l.i0], an[l.i1]);
                c = sn * a;
                if (phi_out != nullptr && u == phi_col) phi_out[(size_t)n * g.H + h] = asinf(sn);
            }
            tile[i][lane] = c;
        }
        __syncthreads();
        if (lane < nb) {
            float s = 0.0f;
            for (int hh = 0; hh < g.H; ++hh) s += tile[lane][hh];
            wave[(size_t)n * g.Lw + u0 + b0 + lane] = s / (float)g.H;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" size_t alive_oscillator_workspace_bytes(int N, int H, int Lf) {
    return align_up((size_t)N * H * Lf * sizeof(double), 256) + align_up((size_t)N * H * sizeof(float), 256);
}

extern "C" int alive_oscillator(const float* amps, const float* f0, const float* phi_in, int N, int H, int Lf, int seg,
                                float sample_rate, int crop0, int phi_col, float* wave, float* phi_out, void* ws,
                                void* stream) {
    ALIVE_CHECK_ARG(amps && f0 && wave && ws, "alive_oscillator: null pointer");
    ALIVE_CHECK_ARG(N > 0 && H > 0 && H <= 64 && Lf > 0 && seg > 0, "alive_oscillator: bad sizes (H <= 64)");
    const int Lw = Lf * seg;
    ALIVE_CHECK_ARG(crop0 >= 0 && crop0 < Lw, "alive_oscillator: crop0 %d outside [0,%d)", crop0, Lw);
    ALIVE_CHECK_ARG(phi_out == nullptr || (phi_col >= 0 && phi_col < Lw), "alive_oscillator: phi_col outside wave");
    OscGeom g{H, Lf, seg, Lw, (float)Lf / (float)Lw, sample_rate};
    Arena a(ws);
    double* S = a.take<double>((size_t)N * H * Lf);
    float* dt0 = a.take<float>((size_t)N * H);
    hipStream_t s = (hipStream_t)stream;
    osc_segsum_kernel<<<dim3(Lf, N, cdiv(H, 64)), 64, 0, s>>>(f0, g, S);
    osc_prefix_kernel<<<dim3(N, cdiv(H, 64)), 64, 0, s>>>(f0, g, crop0, S, dt0);
    osc_synth_kernel<<<dim3(Lf, N), 64, 0, s>>>(amps, f0, phi_in, g, S, dt0, phi_col, wave, phi_out);
    ALIVE_CHECK_LAUNCH("alive_oscillator");
    return ALIVE_OK;
}
