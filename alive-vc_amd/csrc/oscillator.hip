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
    int Q, sub;         // a frame's seg samples are walked as Q pieces of sub samples, one wave each (Q = 1: whole frames)
};

// Few frames (the streaming step: one window of 8): a wave per frame leaves 8 waves on the chip, each behind a serial chain of
// 320 samples (fp64 add -> fp32 phase -> argument reduction in fp64 -> sin: ~500 cycles per sample).  The chain only carries the
// running sum, so a frame is cut into Q pieces whose starting sums come from the same segment-sum / prefix pass at piece
// granularity -- the fp64 partial sums are re-associated exactly like the per-frame sums already are.
__host__ __device__ inline int osc_pieces(int N, int Lf, int seg) { return ((int64_t)N * Lf <= 64 && seg % 8 == 0) ? 8 : 1; }

// x / 16000 correctly rounded, as one multiply and two fmas: q = x * rc, then one residual correction.  For this
// divisor the result equals the IEEE quotient for EVERY fp32 x in [2^-24, 2^24) -- checked exhaustively on the GPU
// (tests/test_gpu_ops.py::test_div16000_is_the_ieee_quotient); other sample rates take the IEEE sequence.
__device__ __forceinline__ float div_rate(float x, float sample_rate) {
    if (sample_rate == 16000.0f) {
        const float rc = 6.25e-5f;
        const float q = x * rc;
        const float r = fmaf(-q, 16000.0f, x);
        return fmaf(r, rc, q);
    }
    return __fdiv_rn(x, sample_rate);
}

__device__ __forceinline__ float formant_step(const float* f0n, float hmul, const Lerp& l, float sample_rate) {
    float a = f0n[l.i0] * hmul;
    float b = f0n[l.i1] * hmul;
    return div_rate(lerp_apply(l, a, b), sample_rate);
}

// sin of an fp32 phase that reaches 1e6 rad: the argument is reduced in fp64 (k = rint(x * 2/pi), r = x - k * pi/2 with a
// two-term pi/2: exact to 1e-10 rad at these magnitudes), then the Cephes single-precision sin / cos kernels on
// |r| <= pi/4 (< 2 ulp).  ocml's sinf takes its Payne-Hanek path for such arguments: ~70 instructions against ~25.
__device__ __forceinline__ float sin_phase(float theta) {
    const double t = (double)theta;
    const double k = rint(t * 0.63661977236758134308);
    double r = fma(-k, 1.57079632679489655800, t);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const int q = (int)k;
    const float x = (float)r, x2 = x * x;
    float sp = fmaf(-1.9515295891e-4f, x2, 8.3321608736e-3f);
    sp = fmaf(sp, x2, -1.6666654611e-1f);
    const float sn = fmaf(sp * x2, x, x);
    float cp = fmaf(2.443315711809948e-5f, x2, -1.388731625493765e-3f);
    cp = fmaf(cp, x2, 4.166664568298827e-2f);
    const float cs = fmaf(cp * x2, x2, fmaf(-0.5f, x2, 1.0f));
    const float v = (q & 1) ? cs : sn;
    return (q & 2) ? -v : v;
}

__global__ __launch_bounds__(64) void osc_segsum_kernel(const float* __restrict__ f0, OscGeom g, double* __restrict__ S) {
    const int fq = blockIdx.x, n = blockIdx.y;               // piece fq = f * Q + q
    const int h = blockIdx.z * 64 + threadIdx.x;
    if (h >= g.H) return;
    const float* f0n = f0 + (size_t)n * g.Lf;
    const float hmul = (float)(h + 1);
    double acc = 0.0;
    const int u0 = fq * g.sub;
    for (int i = 0; i < g.sub; ++i) {
        Lerp l = lerp_coord(u0 + i, g.ratio, g.Lf);
        acc += (double)formant_step(f0n, hmul, l, g.sample_rate);
    }
    S[((size_t)n * g.H + h) * g.Lf * g.Q + fq] = acc;
}

// exclusive scan of the segment sums (in place) and dt0 = fp32 prefix at crop0
__global__ __launch_bounds__(64) void osc_prefix_kernel(const float* __restrict__ f0, OscGeom g, int crop0,
                                                        double* __restrict__ S, float* __restrict__ dt0) {
    const int n = blockIdx.x;
    const int h = blockIdx.y * 64 + threadIdx.x;
    if (h >= g.H) return;
    double* s = S + ((size_t)n * g.H + h) * g.Lf * g.Q;
    double run = 0.0;
    const int fc = crop0 / g.sub;                            // the piece crop0 lies in
    double at_fc = 0.0;
    for (int f = 0; f < g.Lf * g.Q; ++f) {
        double v = s[f];
        s[f] = run;
        if (f == fc) at_fc = run;
        run += v;
    }
    const float* f0n = f0 + (size_t)n * g.Lf;
    const float hmul = (float)(h + 1);
    double acc = at_fc;
    for (int u = fc * g.sub; u <= crop0; ++u) {
        Lerp l = lerp_coord(u, g.ratio, g.Lf);
        acc += (double)formant_step(f0n, hmul, l, g.sample_rate);
    }
    dt0[(size_t)n * g.H + h] = (float)acc;
}

constexpr int MAX_SEG = 512;
// samples per transpose batch of osc_synth_kernel.  The [BT][65] tile is what limits the blocks (one wave each, a serial fp64 phase
// chain inside) per CU: 64 -> 16 takes 128 windows from 3.06 to 2.09 ms, bit for bit the same wave (8: the same, 4: slower)
constexpr int OSC_BT = 16;

__global__ __launch_bounds__(64) void osc_synth_kernel(const float* __restrict__ amps, const float* __restrict__ f0,
                                                       const float* __restrict__ phi_in, OscGeom g,
                                                       const double* __restrict__ P, const float* __restrict__ dt0,
                                                       int phi_col, int f_off, int amp_ld, float* __restrict__ wave,
                                                       float* __restrict__ phi_out) {
    constexpr int BT = OSC_BT;
    __shared__ float tile[BT][65];
    __shared__ uint2 coord[MAX_SEG];              // per sample of the frame: (i0 | i1 << 16, w1)
    // frame f of the WINDOW; amps / wave hold the frames [f_off, f_off + amp_ld) only (range mode; else f_off = 0, amp_ld = Lf)
    const int f = blockIdx.x / g.Q + f_off, q = blockIdx.x % g.Q, n = blockIdx.y;      // piece q of frame f
    const int lane = threadIdx.x;
    const int h = lane;
    const bool hv = h < g.H;
    const float* f0n = f0 + (size_t)n * g.Lf;
    const float* an = amps + ((size_t)n * g.H + (hv ? h : 0)) * amp_ld;
    const float hmul = (float)(h + 1);
    const float TWO_PI_F = 6.283185307179586f;
    const int u0 = f * g.seg + q * g.sub;
    for (int i = lane; i < g.sub; i += 64) {
        const Lerp l = lerp_coord(u0 + i, g.ratio, g.Lf);
        coord[i] = make_uint2((unsigned)l.i0 | ((unsigned)l.i1 << 16), __float_as_uint(l.w1));
    }
    // the interpolation of a frame's samples touches the frames f - 1, f, f + 1 only: keep their values in registers
    const int fm = f > 0 ? f - 1 : 0, fp = f + 1 < g.Lf ? f + 1 : g.Lf - 1;
    const float fo_m = f0n[fm] * hmul, fo_c = f0n[f] * hmul, fo_p = f0n[fp] * hmul;
    auto acol = [&](int fr) { int c = fr - f_off; return c < 0 ? 0 : (c < amp_ld ? c : amp_ld - 1); };
    const float am_m = an[acol(fm)], am_c = an[acol(f)], am_p = an[acol(fp)];
    double acc = hv ? P[((size_t)n * g.H + h) * g.Lf * g.Q + f * g.Q + q] : 0.0;
    const float d0 = hv ? dt0[(size_t)n * g.H + h] : 0.0f;
    const float ph = (hv && phi_in != nullptr) ? phi_in[(size_t)n * g.H + h] : 0.0f;
    __syncthreads();
    for (int b0 = 0; b0 < g.sub; b0 += BT) {
        const int nb = (g.sub - b0) < BT ? (g.sub - b0) : BT;
        // The (i0, i1) pair of F.interpolate changes once inside a frame (at its midpoint) and the batches of BT samples do not
        // straddle it when seg / 2 is a multiple of BT (320 / 2 = 160): a batch whose first and last sample share the pair takes
        // its four operands ONCE instead of selecting them per sample (12 compares / selects of ~50 instruction slots per element;
        // same values, same arithmetic).  Wave-uniform test: the table is the same for every lane.
        const unsigned pair0 = __builtin_amdgcn_readfirstlane(coord[b0].x), pair1 = __builtin_amdgcn_readfirstlane(coord[b0 + nb - 1].x);
        const bool same = pair0 == pair1;
        const int j0 = pair0 & 0xffff, j1 = pair0 >> 16;
        const float fa_b = j0 == f ? fo_c : (j0 < f ? fo_m : fo_p), fb_b = j1 == f ? fo_c : (j1 < f ? fo_m : fo_p);
        const float aa_b = j0 == f ? am_c : (j0 < f ? am_m : am_p), ab_b = j1 == f ? am_c : (j1 < f ? am_m : am_p);
        for (int i = 0; i < nb; ++i) {
            const uint2 xc = coord[b0 + i];
            const float w1 = __uint_as_float(xc.y), w0 = 1.0f - w1;
            float fa = fa_b, fb = fb_b, aa = aa_b, ab = ab_b;
            if (!same) {                                       // wave-uniform branch: the batch that holds a change of the pair
                const int i0 = xc.x & 0xffff, i1 = xc.x >> 16;
                fa = i0 == f ? fo_c : (i0 < f ? fo_m : fo_p); fb = i1 == f ? fo_c : (i1 < f ? fo_m : fo_p);
                aa = i0 == f ? am_c : (i0 < f ? am_m : am_p); ab = i1 == f ? am_c : (i1 < f ? am_m : am_p);
            }
            float c = 0.0f;
            if (hv) {
                acc += (double)div_rate(fmaf(w0, fa, w1 * fb), g.sample_rate);
                const float dt = (float)acc - d0;
                const float theta = __fadd_rn(__fmul_rn(TWO_PI_F, dt), ph);
                const float sn = sin_phase(theta);
                c = sn * fmaf(w0, aa, w1 * ab);
                if (phi_out != nullptr && u0 + b0 + i == phi_col) phi_out[(size_t)n * g.H + h] = asinf(sinf(theta));
            }
            tile[i][lane] = c;
        }
        __syncthreads();
        if (lane < nb) {
            float s = 0.0f;
            for (int hh = 0; hh < g.H; ++hh) s += tile[lane][hh];
            wave[(size_t)n * amp_ld * g.seg + (u0 - f_off * g.seg) + b0 + lane] = s / (float)g.H;
        }
        __syncthreads();
    }
}

__global__ void div16000_check_kernel(unsigned first, unsigned count, unsigned* __restrict__ mismatches) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float x = __uint_as_float(first + i);
    if (__float_as_uint(div_rate(x, 16000.0f)) != __float_as_uint(__fdiv_rn(x, 16000.0f))) atomicAdd(mismatches, 1u);
}

}  // namespace

extern "C" size_t alive_oscillator_workspace_bytes(int N, int H, int Lf) {
    const int Q = osc_pieces(N, Lf, 8);           // (the piece count the launch may pick, whatever the segment length)
    return align_up((size_t)N * H * Lf * Q * sizeof(double), 256) + align_up((size_t)N * H * sizeof(float), 256);
}

extern "C" int alive_oscillator(const float* amps, const float* f0, const float* phi_in, int N, int H, int Lf, int seg,
                                float sample_rate, int crop0, int phi_col, float* wave, float* phi_out, void* ws,
                                void* stream) {
    return alive_oscillator_range(amps, f0, phi_in, N, H, Lf, seg, sample_rate, crop0, phi_col, 0, Lf, wave, phi_out, ws, stream);
}

extern "C" int alive_oscillator_range(const float* amps, const float* f0, const float* phi_in, int N, int H, int Lf, int seg,
                                      float sample_rate, int crop0, int phi_col, int f_begin, int n_frames, float* wave,
                                      float* phi_out, void* ws, void* stream) {
    ALIVE_CHECK_ARG(amps && f0 && wave && ws, "alive_oscillator: null pointer");
    ALIVE_CHECK_ARG(f_begin >= 0 && n_frames > 0 && f_begin + n_frames <= Lf, "alive_oscillator: frame range [%d, %d) outside [0, %d)",
                    f_begin, f_begin + n_frames, Lf);
    ALIVE_CHECK_ARG(N > 0 && H > 0 && H <= 64 && Lf > 0 && Lf < 65536 && seg > 0 && seg <= MAX_SEG,
                    "alive_oscillator: bad sizes (H <= 64, seg <= %d)", MAX_SEG);
    const int Lw = Lf * seg;
    ALIVE_CHECK_ARG(crop0 >= 0 && crop0 < Lw, "alive_oscillator: crop0 %d outside [0,%d)", crop0, Lw);
    ALIVE_CHECK_ARG(phi_out == nullptr || (phi_col >= 0 && phi_col < Lw), "alive_oscillator: phi_col outside wave");
    const int Q = osc_pieces(N, Lf, seg);
    OscGeom g{H, Lf, seg, Lw, (float)Lf / (float)Lw, sample_rate, Q, seg / Q};
    Arena a(ws);
    double* S = a.take<double>((size_t)N * H * Lf * Q);
    float* dt0 = a.take<float>((size_t)N * H);
    hipStream_t s = (hipStream_t)stream;
    osc_segsum_kernel<<<dim3(Lf * Q, N, cdiv(H, 64)), 64, 0, s>>>(f0, g, S);
    osc_prefix_kernel<<<dim3(N, cdiv(H, 64)), 64, 0, s>>>(f0, g, crop0, S, dt0);
    osc_synth_kernel<<<dim3(n_frames * Q, N), 64, 0, s>>>(amps, f0, phi_in, g, S, dt0, phi_col, f_begin, n_frames, wave, phi_out);
    ALIVE_CHECK_LAUNCH("alive_oscillator");
    return ALIVE_OK;
}

// test hook: counts the fp32 bit patterns in [first, first + count) for which the fast division by 16000 differs from
// the IEEE quotient (must be 0 over the whole range the oscillator can see)
extern "C" int alive_debug_div16000_mismatches(unsigned first, unsigned count, unsigned* mismatches, void* stream) {
    ALIVE_CHECK_ARG(mismatches && count > 0, "alive_debug_div16000_mismatches: bad args");
    div16000_check_kernel<<<(count + 255) / 256, 256, 0, (hipStream_t)stream>>>(first, count, mismatches);
    ALIVE_CHECK_LAUNCH("alive_debug_div16000_mismatches");
    return ALIVE_OK;
}
