// Shared epilogue of the conv GEMM kernels (conv.hip: f32 MFMA, conv_split.hip: split-bf16 MFMA):
// activation -> +post_add -> *ch_scale -> +residual -> +skip -> raw store, optional second output
// Z = gelu(v) * interp(film scale) + interp(film shift)   (decoder.py:112-117,130-132), or the
// ConvTranspose1d(k == stride) scatter store.
#pragma once
#include "common.h"

// Kept out of line: dozens of inlined copies of erff/expf/sinf (one per accumulator element) blow the
// unrolled epilogue past the point where the accumulators stay in registers.
__device__ __attribute__((noinline)) static float apply_act(float v, int act) {
    if (act == 1) return gelu_erf(v);
    if (act == 2) return expf(v);
    return sinf(v);
}

__device__ __forceinline__ void conv_epilogue_store(const AliveConv& p, int n, int row, int t, float v, const Lerp& lp) {
    if (p.act != 0) v = apply_act(v, p.act);
    if (p.post_add != nullptr) v = v + p.post_add[row];
    if (p.ch_scale != nullptr) v = v * p.ch_scale[row];
    if (p.up == 1) {
        const size_t o = ((size_t)n * p.Co + row) * p.Tout + t;
        if (p.residual != nullptr) v = v + p.residual[o];
        if (p.skip != nullptr) v = v + p.skip[o];
        if (p.Y != nullptr) p.Y[o] = v;
        if (p.Z != nullptr) {
            const float* fs = p.film + ((size_t)n * p.film_rows + p.film_scale_row + row) * p.Lf;
            const float* fh = p.film + ((size_t)n * p.film_rows + p.film_shift_row + row) * p.Lf;
            float sc = lerp_apply(lp, fs[lp.i0], fs[lp.i1]);
            float sh = lerp_apply(lp, fh[lp.i0], fh[lp.i1]);
            float g = apply_act(v, 1);
            p.Z[o] = g * sc + sh;
        }
    } else {
        const int co = row / p.up, jj = row - co * p.up;
        p.Y[((size_t)n * (p.Co / p.up) + co) * ((size_t)p.Tout * p.up) + (size_t)t * p.up + jj] = v;
    }
}
