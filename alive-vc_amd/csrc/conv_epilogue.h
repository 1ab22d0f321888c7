// Shared epilogue of the conv GEMM kernels (conv.hip: f32 MFMA, conv_split.hip: split-bf16 MFMA):
// activation -> +post_add -> *ch_scale -> +residual -> +skip -> raw store, optional second output
// Z = gelu(v) * interp(film scale) + interp(film shift)   (decoder.py:112-117,130-132), or the
// ConvTranspose1d(k == stride) scatter store.
#pragma once
#include "common.h"

// exact activations, kept out of line: dozens of inlined copies of erff/expf/sinf (one per accumulator
// element) blow the unrolled epilogue past the point where the accumulators stay in registers.
__device__ __attribute__((noinline)) static float apply_act(float v, int act) {
    if (act == 1) return gelu_erf(v);
    if (act == 2) return expf(v);
    return sinf(v);
}

// Branch-free GELU for the split-bf16 paths: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7) on v_rcp_f32 / v_exp_f32
// (the hardware reciprocal directly: __frcp_rn expands to the 12-instruction IEEE division sequence); ~17 VALU, inlined.
// Only used next to the split-bf16 GEMMs and the FiLM second output, whose own error is at or above it.
__device__ __forceinline__ float gelu_fast(float x) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p = p * t;
    const float e = __expf(-ax * ax);
    const float erf_abs = fmaf(-p, e, 1.0f);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// the same on two values per lane: the polynomial runs on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 ax = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
    const f32x2 d = pk_fma(pk_splat(0.3275911f), ax, pk_splat(1.0f));
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 p = pk_fma(pk_splat(1.061405429f), t, pk_splat(-1.453152027f));
    p = pk_fma(p, t, pk_splat(1.421413741f));
    p = pk_fma(p, t, pk_splat(-0.284496736f));
    p = pk_fma(p, t, pk_splat(0.254829592f));
    p = p * t;
    const f32x2 m = -ax * ax;
    const f32x2 e = {__expf(m[0]), __expf(m[1])};
    const f32x2 erf_abs = pk_fma(-p, e, pk_splat(1.0f));
    const f32x2 s = {copysignf(erf_abs[0], x[0]), copysignf(erf_abs[1], x[1])};
    return (x * 0.5f) * (s + 1.0f);
}

// FiLM slab of one block in LDS: [row_local][2 (scale, shift)][FILM_NF frames]
constexpr int FILM_NF = 20;

struct FilmTile {
    const float* lds;     // nullptr -> read film from global memory
    int f_lo;
};

// frames [f_lo, f_lo + nf) cover every interpolation tap of output columns [t0, t0 + ncols)
__device__ __forceinline__ bool film_tile_range(const AliveConv& p, float ratio, int t0, int ncols, int& f_lo, int& nf) {
    int t1 = t0 + ncols - 1;
    if (t1 > p.Tout - 1) t1 = p.Tout - 1;
    Lerp a = lerp_coord(t0 + p.film_t0, ratio, p.Lf), b = lerp_coord(t1 + p.film_t0, ratio, p.Lf);
    f_lo = a.i0;
    nf = b.i1 - a.i0 + 1;
    return nf <= FILM_NF;
}

// cooperative load by the whole block (256 threads); call between two __syncthreads()
__device__ __forceinline__ void film_tile_load(const AliveConv& p, int n, int m0, int rows, int f_lo, int nf, float* lds) {
    for (int e = threadIdx.x; e < rows * 2 * FILM_NF; e += 256) {
        int f = e % FILM_NF, sel = (e / FILM_NF) & 1, r = e / (2 * FILM_NF);
        int row = m0 + r;
        float v = 0.0f;
        if (row < p.Co && f < nf) {
            int base = sel == 0 ? p.film_scale_row : p.film_shift_row;
            v = p.film[((size_t)n * p.film_rows + base + row) * p.Lf + f_lo + f];
        }
        lds[e] = v;
    }
}

template <bool FAST>
__device__ __forceinline__ void conv_epilogue_store(const AliveConv& p, int n, int row, int row_local, int t, float v,
                                                    const Lerp& lp, const FilmTile& ft) {
    if (FAST) {               // split-bf16 kernel: everything inline (a call would spill the accumulators); no sin
        if (p.act == 1) v = gelu_fast(v);
        else if (p.act == 2) v = expf(v);
    } else if (p.act != 0) {
        v = apply_act(v, p.act);
    }
    if (p.post_add != nullptr) v = v + p.post_add[row];
    if (p.ch_scale != nullptr) v = v * p.ch_scale[row];
    if (p.up == 1) {
        const size_t o = ((size_t)n * p.Co + row) * p.Tout + t;
        if (p.residual != nullptr) v = v + p.residual[o];
        if (p.skip != nullptr) v = v + p.skip[o];
        if (p.Y != nullptr) p.Y[o] = v;
        if (p.Z != nullptr) {
            float s0, s1, h0, h1;
            if (ft.lds != nullptr) {
                const float* fs = ft.lds + row_local * 2 * FILM_NF - ft.f_lo;
                s0 = fs[lp.i0]; s1 = fs[lp.i1]; h0 = fs[FILM_NF + lp.i0]; h1 = fs[FILM_NF + lp.i1];
            } else {
                const float* fs = p.film + ((size_t)n * p.film_rows + p.film_scale_row + row) * p.Lf;
                const float* fh = p.film + ((size_t)n * p.film_rows + p.film_shift_row + row) * p.Lf;
                s0 = fs[lp.i0]; s1 = fs[lp.i1]; h0 = fh[lp.i0]; h1 = fh[lp.i1];
            }
            float sc = lerp_apply(lp, s0, s1);
            float sh = lerp_apply(lp, h0, h1);
            float g = gelu_fast(v);
            p.Z[o] = g * sc + sh;
        }
    } else {
        const int co = row / p.up, jj = row - co * p.up;
        p.Y[((size_t)n * (p.Co / p.up) + co) * ((size_t)p.Tout * p.up) + (size_t)t * p.up + jj] = v;
    }
}
