// Conv1d (stride 1) / ConvTranspose1d(k == stride) as an implicit GEMM on the bf16 MFMA with
// 2-term split operands ("bf16x3"): every fp32 value v is carried as hi = bf16(v), lo = bf16(v - hi) and
//      a*b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (fp32 accumulate in the MFMA)
// which keeps 16 mantissa bits per operand: relative error ~2^-16 per product instead of bf16's 2^-9,
// at 3 bf16 MFMAs per product = 16/3 the rate of the f32-input MFMA.  Measured on the whole decoder
// (oracle simulation): waveform RMS error 3.5e-6 against the 1e-3 bar.  Used for the
// decoder's GEMMs only; the content encoder / f0 estimator (top-k / argmax downstream) stay on the exact
// f32 MFMA kernel of conv.hip.
//
//   weights  W16[2][Co_pad][KW*Ci_pad] bf16, tap-major k = j*Ci_pad + ci (packed by module/_pack.py)
//   X        fp32 [N][Ci][Tin]; split on the fly while staging
//
// A k-step is (ci-block of 32 channels, tap j).  The X tile of a ci-block is staged ONCE, transposed to
// [time + halo][ci] so that the 8 consecutive k of an MFMA B fragment are contiguous, and all KW taps read
// it at shifted rows -- the im2col replication never exists, not even in LDS.  Weights tiles are double
// buffered per k-step.  Block = 4 waves, tile BM x 128 (BM = 128: 2x2 waves of 64x64; BM = 64: 1x4 waves
// of 64x32), v_mfma_f32_32x32x16_bf16.
#include "conv_epilogue.h"

namespace {

constexpr int BN = 128;
constexpr int BKC = 32;           // channels per ci-block
constexpr int XROWS = 144;        // BN + max halo (4 taps x dilation 4)
constexpr int PITCH = 80;         // bytes per LDS row (32 bf16 + 16 B pad): 16-B aligned rows, conflict-free ds_read_b128
constexpr int XPLANE = XROWS * PITCH;

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p) { return *(const bf16x8*)p; }

template <int BM>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(AliveConv p, float film_ratio) {
    constexpr int WN = BM == 128 ? 2 : 4;         // waves along time
    constexpr int TN = BN / WN;                   // 64 or 32
    constexpr int NR = TN / 32;                   // 2 or 1
    constexpr int MR = 2;                         // 64 rows per wave
    constexpr int APLANE = BM * PITCH;
    constexpr int A_ITEMS = 2 * BM * 4 / 256;     // 16-B chunks per thread per k-step (4 or 2)

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * XPLANE + 2 * 2 * APLANE];
    unsigned char* Xs = smem;
    unsigned char* As = smem + 2 * XPLANE;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = BM == 128 ? (wid >> 1) : 0, wn = BM == 128 ? (wid & 1) : wid;
    const int lr = lane & 31, lh = lane >> 5;
    const int n = blockIdx.z, m0 = blockIdx.y * BM, t0 = blockIdx.x * BN;
    const int co_pad = (p.Co + 15) & ~15;
    const int K2 = p.KW * p.Ci_pad;
    const int ncb = p.Ci_pad / BKC;
    const int nsteps = ncb * p.KW;
    const int xrows = BN + (p.KW - 1) * p.dil;
    const unsigned short* W16 = (const unsigned short*)p.W;
    const float* Xn = p.X + (size_t)n * p.Ci * p.Tin;

    // ---- staging registers ----
    u32x4 a_reg[A_ITEMS];
    float x_reg[9][2];

    auto load_A = [&](int step) {
        const int cb = step / p.KW, j = step - cb * p.KW;
        const int kcol = j * p.Ci_pad + cb * BKC;
#pragma unroll
        for (int v = 0; v < A_ITEMS; ++v) {
            int i = v * 256 + tid;
            int c = i & 3, r = (i >> 2) % BM, pl = i / (4 * BM);
            int grow = m0 + r;
            grow = grow < co_pad ? grow : co_pad - 1;
            a_reg[v] = *(const u32x4*)(W16 + ((size_t)pl * co_pad + grow) * K2 + kcol + c * 8);
        }
    };
    auto store_A = [&](int buf) {
#pragma unroll
        for (int v = 0; v < A_ITEMS; ++v) {
            int i = v * 256 + tid;
            int c = i & 3, r = (i >> 2) % BM, pl = i / (4 * BM);
            *(u32x4*)(As + (buf * 2 + pl) * APLANE + r * PITCH + c * 16) = a_reg[v];
        }
    };
    auto load_X = [&](int cb) {
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            int i = it * 256 + tid;
            int r = i % XROWS, pair = i / XROWS;
            int tin = t0 - p.pad_left + r;
            if (tin < 0 && p.pad_mode != 0) tin = -tin;
            bool ok = r < xrows && tin >= 0 && tin < p.Tin;
            int ci = cb * BKC + pair * 2;
            x_reg[it][0] = (ok && ci < p.Ci) ? Xn[(size_t)ci * p.Tin + tin] : 0.0f;
            x_reg[it][1] = (ok && ci + 1 < p.Ci) ? Xn[(size_t)(ci + 1) * p.Tin + tin] : 0.0f;
        }
    };
    auto store_X = [&]() {
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            int i = it * 256 + tid;
            int r = i % XROWS, pair = i / XROWS;
            float x0 = x_reg[it][0], x1 = x_reg[it][1];
            unsigned h0 = f32_to_bf16_rn(x0), h1 = f32_to_bf16_rn(x1);
            unsigned l0 = f32_to_bf16_rn(x0 - __uint_as_float(h0 << 16));
            unsigned l1 = f32_to_bf16_rn(x1 - __uint_as_float(h1 << 16));
            *(unsigned*)(Xs + r * PITCH + pair * 4) = h0 | (h1 << 16);
            *(unsigned*)(Xs + XPLANE + r * PITCH + pair * 4) = l0 | (l1 << 16);
        }
    };

    // ---- accumulators start at the bias ----
    f32x16 acc[MR][NR];
#pragma unroll
    for (int mm = 0; mm < MR; ++mm) {
        f32x16 b16;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = m0 + wm * 64 + mm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            b16[r] = (p.bias != nullptr && row < p.Co) ? p.bias[row] : 0.0f;
        }
#pragma unroll
        for (int nn = 0; nn < NR; ++nn) acc[mm][nn] = b16;
    }

    load_A(0);
    load_X(0);
    store_A(0);
    store_X();
    __syncthreads();
    int cb = 0, j = 0;
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const bool more = step + 1 < nsteps;
        const bool next_cb = more && (j + 1 == p.KW);
        if (more) load_A(step + 1);
        if (next_cb) load_X(cb + 1);

        const unsigned char* Ab = As + buf * 2 * APLANE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 ah[MR], al[MR], bh[NR], bl[NR];
#pragma unroll
            for (int mm = 0; mm < MR; ++mm) {
                const unsigned char* a = Ab + (wm * 64 + mm * 32 + lr) * PITCH + s * 32 + lh * 16;
                ah[mm] = lds_frag(a);
                al[mm] = lds_frag(a + APLANE);
            }
#pragma unroll
            for (int nn = 0; nn < NR; ++nn) {
                const unsigned char* b = Xs + (wn * TN + nn * 32 + lr + j * p.dil) * PITCH + s * 32 + lh * 16;
                bh[nn] = lds_frag(b);
                bl[nn] = lds_frag(b + XPLANE);
            }
#pragma unroll
            for (int mm = 0; mm < MR; ++mm)
#pragma unroll
                for (int nn = 0; nn < NR; ++nn) {
                    acc[mm][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mm], bh[nn], acc[mm][nn], 0, 0, 0);
                    acc[mm][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mm], bl[nn], acc[mm][nn], 0, 0, 0);
                    acc[mm][nn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mm], bh[nn], acc[mm][nn], 0, 0, 0);
                }
        }
        if (next_cb) {
            __syncthreads();                // every wave is done with the X tile of this ci-block
            store_X();
        }
        if (more) store_A(buf ^ 1);
        __syncthreads();
        if (++j == p.KW) { j = 0; ++cb; }
    }

    // ---- epilogue: accumulators -> LDS -> cooperative row-wise pass ----
    // Staging the tile through LDS turns the MFMA layout (a lane owns 16 scattered rows of one column) into
    // whole rows: every global access of the epilogue (residual, skip, Y, Z) is a 16-B vector per thread on
    // 512-B contiguous row segments, and the element code exists once in a rolled loop instead of 64 times.
    constexpr int PR = BM == 128 ? 64 : 32;          // rows per pass (two passes: mm = 0, 1)
    constexpr int CP = BN + 4;                       // fp32 pitch of the staged tile
    float* Ct = (float*)smem;                        // [PR][CP]
    float* Ft = Ct + PR * CP;                        // [PR][2][FILM_NF]
    static_assert((PR * CP + PR * 2 * FILM_NF) * 4 <= (int)sizeof(smem), "epilogue staging fits");
    int f_lo = 0, nf = 0;
    if (p.Z != nullptr) film_tile_range(p, film_ratio, t0, BN, f_lo, nf);       // fit is checked on the host
    const bool vec = (p.up == 1) && ((p.Tout & 3) == 0);
#pragma unroll
    for (int mm = 0; mm < MR; ++mm) {
#pragma unroll
        for (int nn = 0; nn < NR; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ct[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * CP + wn * TN + nn * 32 + lr] = acc[mm][nn][r];
        if (p.Z != nullptr) {
            for (int e = tid; e < PR * 2 * FILM_NF; e += 256) {
                int f = e % FILM_NF, sel = (e / FILM_NF) & 1, pr = e / (2 * FILM_NF);
                int row = m0 + (pr >> 5) * 64 + mm * 32 + (pr & 31);
                float v = 0.0f;
                if (row < p.Co && f < nf)
                    v = p.film[((size_t)n * p.film_rows + (sel == 0 ? p.film_scale_row : p.film_shift_row) + row) * p.Lf + f_lo + f];
                Ft[e] = v;
            }
        }
        __syncthreads();
        if (vec) {
#pragma unroll 1
            for (int g = tid; g < PR * (BN / 4); g += 256) {
                const int pr = g >> 5, c4 = (g & 31) * 4;
                const int row = m0 + (pr >> 5) * 64 + mm * 32 + (pr & 31);
                const int t = t0 + c4;
                if (row >= p.Co || t >= p.Tout) continue;
                f32x4 v = *(const f32x4*)&Ct[pr * CP + c4];
                if (p.act == 1) { for (int q = 0; q < 4; ++q) v[q] = gelu_fast(v[q]); }
                else if (p.act == 2) { for (int q = 0; q < 4; ++q) v[q] = expf(v[q]); }
                if (p.post_add != nullptr) v = v + p.post_add[row];
                if (p.ch_scale != nullptr) v = v * p.ch_scale[row];
                const size_t o = ((size_t)n * p.Co + row) * p.Tout + t;
                if (p.residual != nullptr) v = v + *(const f32x4*)(p.residual + o);
                if (p.skip != nullptr) v = v + *(const f32x4*)(p.skip + o);
                if (p.Y != nullptr) *(f32x4*)(p.Y + o) = v;
                if (p.Z != nullptr) {
                    const float* fs = Ft + pr * 2 * FILM_NF - f_lo;
                    f32x4 z;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        Lerp lp = lerp_coord(t + q, film_ratio, p.Lf);
                        float sc = lerp_apply(lp, fs[lp.i0], fs[lp.i1]);
                        float sh = lerp_apply(lp, fs[FILM_NF + lp.i0], fs[FILM_NF + lp.i1]);
                        z[q] = gelu_fast(v[q]) * sc + sh;
                    }
                    *(f32x4*)(p.Z + o) = z;
                }
            }
        } else {
#pragma unroll 1
            for (int g = tid; g < PR * BN; g += 256) {
                const int pr = g >> 7, c = g & 127;
                const int row = m0 + (pr >> 5) * 64 + mm * 32 + (pr & 31);
                const int t = t0 + c;
                if (row >= p.Co || t >= p.Tout) continue;
                float v = Ct[pr * CP + c];
                if (p.act == 1) v = gelu_fast(v);
                else if (p.act == 2) v = expf(v);
                if (p.post_add != nullptr) v = v + p.post_add[row];
                if (p.ch_scale != nullptr) v = v * p.ch_scale[row];
                if (p.up == 1) {
                    const size_t o = ((size_t)n * p.Co + row) * p.Tout + t;
                    if (p.residual != nullptr) v = v + p.residual[o];
                    if (p.skip != nullptr) v = v + p.skip[o];
                    if (p.Y != nullptr) p.Y[o] = v;
                    if (p.Z != nullptr) {
                        const float* fs = Ft + pr * 2 * FILM_NF - f_lo;
                        Lerp lp = lerp_coord(t, film_ratio, p.Lf);
                        float sc = lerp_apply(lp, fs[lp.i0], fs[lp.i1]);
                        float sh = lerp_apply(lp, fs[FILM_NF + lp.i0], fs[FILM_NF + lp.i1]);
                        p.Z[o] = gelu_fast(v) * sc + sh;
                    }
                } else {
                    const int co = row / p.up, jj = row - co * p.up;
                    p.Y[((size_t)n * (p.Co / p.up) + co) * ((size_t)p.Tout * p.up) + (size_t)t * p.up + jj] = v;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

int alive_conv_split_launch(const AliveConv* d, float ratio, hipStream_t s) {
    ALIVE_CHECK_ARG(d->stride == 1, "alive_conv1d(split): stride must be 1");
    ALIVE_CHECK_ARG(d->Ci_pad % BKC == 0 && d->Ci_pad >= d->Ci, "alive_conv1d(split): Ci_pad %d for Ci %d", d->Ci_pad, d->Ci);
    ALIVE_CHECK_ARG(d->KW <= 8 && (d->KW - 1) * d->dil <= XROWS - BN, "alive_conv1d(split): halo (KW-1)*dil = %d exceeds %d",
                    (d->KW - 1) * d->dil, XROWS - BN);
    ALIVE_CHECK_ARG(d->pad_mode == 0 || d->pad_mode == 1, "alive_conv1d(split): pad_mode");
    ALIVE_CHECK_ARG(((((uintptr_t)d->Y) | ((uintptr_t)d->Z) | ((uintptr_t)d->residual) | ((uintptr_t)d->skip)) & 15) == 0,
                    "alive_conv1d(split): Y / Z / residual / skip must be 16-byte aligned");
    if (d->Z) {
        const double span = (double)d->Lf / (double)d->Tout * BN + 3.0;
        ALIVE_CHECK_ARG(span <= FILM_NF, "alive_conv1d(split): FiLM second output needs Tout >= ~8 Lf (got Lf %d, Tout %d)", d->Lf, d->Tout);
    }
    ALIVE_CHECK_ARG(d->act >= 0 && d->act <= 2, "alive_conv1d(split): activation %d not available on the split kernel", d->act);
    ALIVE_CHECK_ARG(d->Tout <= d->Tin + d->pad_left, "alive_conv1d(split): Tout");
    if (d->Co > 64) {
        dim3 g(cdiv(d->Tout, BN), cdiv(d->Co, 128), d->N);
        conv_split_kernel<128><<<g, 256, 0, s>>>(*d, ratio);
    } else {
        dim3 g(cdiv(d->Tout, BN), 1, d->N);
        conv_split_kernel<64><<<g, 256, 0, s>>>(*d, ratio);
    }
    ALIVE_CHECK_LAUNCH("alive_conv1d(split)");
    return ALIVE_OK;
}
