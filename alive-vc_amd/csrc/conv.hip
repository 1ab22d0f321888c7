// Conv1d / ConvTranspose1d(k == stride) as an implicit GEMM on the f32-input MFMA
// (v_mfma_f32_16x16x4_f32: bit-for-bit a k-ordered fp32 fmaf chain), gfx950.
//
//   Y[n][co][t] = epilogue( bias[co] + sum_{ci,j} W[co][ci*KW + j] * X[n][ci][t*stride + j*dil - pad] )
//
// A operand = packed weights [Co_pad][K_pad] (K contiguous), B operand = the
// im2col view of X, gathered on the fly while staging into LDS (never
// materialised).  The accumulator starts at the bias, so a K == 1 conv is the
// single fma(w, x, b) that ATen's CPU path produces (F0Encoder.c1 feeds sin()).
//
// Block: 256 threads = 4 waves, BK = 16 (four 16x16x4 k-steps).  Each lane
// reads 4 consecutive k of its A row with one ds_read_b128; MFMA step s of
// lane-group q consumes k = 4q + s for both operands, so the k permutation is
// consistent and the sum is complete after the four steps.
//
// Reference call sites replaced: every nn.Conv1d / nn.ConvTranspose1d of
// module/common.py:48-51,88-92, content_encoder.py:15-19, f0_estimator.py:15-20,
// decoder.py:16-17,41,61,108-110,141,164-182.
#include "conv_epilogue.h"
#include "planes_layout.h"

namespace {

constexpr int BK = 16;
constexpr int A_LD = BK + 4;   // 20 floats = 80 B rows: keeps b128 reads 16-B aligned, spreads banks

// YPL: Y is also written as two k-blocked bf16 planes (AliveConv.Yp) -- an instantiation of its own, like UPV
template <int BM, int BN, int WM, int WN, bool UPV = false, bool YPL = false>
__global__ __launch_bounds__(256) void conv_gemm_kernel(AliveConv p, unsigned kw_magic, float film_ratio) {
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MR = TM / 16, NR = TN / 16;
    constexpr int B_LD = BN + 4;
    constexpr int ROWS_PER_PASS = 256 / BN;        // B rows staged per pass of the block
    constexpr int EPT = BK / ROWS_PER_PASS;        // B elements per thread per k-chunk
    constexpr int A_VEC = (BM * 4 + 255) / 256;    // float4 per thread per k-chunk
    static_assert(WM * WN == 4, "4 waves");
    static_assert(256 % BN == 0 || BN == 256, "BN");

    __shared__ __attribute__((aligned(16))) float As[2][BM][A_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][B_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int n = blockIdx.z;
    const int m0 = blockIdx.y * BM;
    const int t0 = blockIdx.x * BN;
    const int K = p.Ci * p.KW;
    const int co_pad = (p.Co + 15) & ~15;
    const int nk = p.K_pad / BK;

    // ---- per-thread B (im2col) coordinates: fixed column, k advances ----
    const int tt = tid % BN;
    const int kk0 = tid / BN;
    const int t_col = t0 + tt;
    const bool col_ok = t_col < p.Tout;
    const int tin_base = t_col * p.stride - p.pad_left;
    const float* Xn = p.X + (size_t)n * p.Ci * p.Tin;

    f32x4 a_reg[A_VEC];
    float b_reg[EPT];

    auto load_global = [&](int kt) {
        const int kbase = kt * BK;
#pragma unroll
        for (int v = 0; v < A_VEC; ++v) {
            int i = tid + v * 256;
            if (BM * 4 >= 256 || i < BM * 4) {
                int row = i >> 2, c4 = i & 3;
                int grow = m0 + row;
                grow = grow < co_pad ? grow : co_pad - 1;
                a_reg[v] = *(const f32x4*)(p.W + (size_t)grow * p.K_pad + kbase + c4 * 4);
            }
        }
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            int k = kbase + kk0 + r * ROWS_PER_PASS;
            int ci, j;
            if (p.KW == 1) { ci = k; j = 0; }
            else if (kw_magic == 0) { ci = 0; j = k; }          // single input channel (DFT frames, source_in)
            else { ci = (int)(((unsigned)k * kw_magic) >> 20); j = k - ci * p.KW; }
            int tin = tin_base + j * p.dil;
            if (tin < 0 && p.pad_mode != 0) tin = -tin;
            if (tin >= p.Tin && p.pad_mode == 2) tin = 2 * (p.Tin - 1) - tin;
            bool ok = col_ok && (k < K) && (tin >= 0) && (tin < p.Tin);
            b_reg[r] = ok ? Xn[(size_t)ci * p.Tin + tin] : 0.0f;
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int v = 0; v < A_VEC; ++v) {
            int i = tid + v * 256;
            if (BM * 4 >= 256 || i < BM * 4) {
                int row = i >> 2, c4 = i & 3;
                *(f32x4*)&As[buf][row][c4 * 4] = a_reg[v];
            }
        }
#pragma unroll
        for (int r = 0; r < EPT; ++r) Bs[buf][kk0 + r * ROWS_PER_PASS][tt] = b_reg[r];
    };

    // ---- accumulators start at the bias ----
    const int lr = lane & 15, lq = lane >> 4;
    f32x4 acc[MR][NR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        f32x4 b4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = m0 + wm * TM + m * 16 + lq * 4 + r;
            b4[r] = (p.bias != nullptr && row < p.Co) ? p.bias[row] : 0.0f;
        }
#pragma unroll
        for (int nn = 0; nn < NR; ++nn) acc[m][nn] = b4;
    }

    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_global(kt + 1);
        f32x4 a[MR];
        float b[NR][4];
#pragma unroll
        for (int m = 0; m < MR; ++m) a[m] = *(const f32x4*)&As[buf][wm * TM + m * 16 + lr][lq * 4];
#pragma unroll
        for (int nn = 0; nn < NR; ++nn)
#pragma unroll
            for (int s = 0; s < 4; ++s) b[nn][s] = Bs[buf][lq * 4 + s][wn * TN + nn * 16 + lr];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int nn = 0; nn < NR; ++nn)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[nn][s], acc[m][nn], 0, 0, 0);
        if (kt + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
    FilmTile ft{nullptr, 0};
    if (p.Z != nullptr) {                    // block-uniform: FiLM rows of this tile -> LDS (reuses the A buffers)
        int f_lo, nf;
        if (film_tile_range(p, film_ratio, t0, BN, f_lo, nf)) {
            static_assert(sizeof(As) >= BM * 2 * FILM_NF * sizeof(float), "film tile fits the A buffers");
            film_tile_load(p, n, m0, BM, f_lo, nf, &As[0][0][0]);
            __syncthreads();
            ft.lds = &As[0][0][0];
            ft.f_lo = f_lo;
        }
    }
    if constexpr (UPV) {      // host: up == 2 or 4, no activation / post_add / ch_scale.  Its own instantiation: as a run-time branch
                              // of the general kernel it slowed the strided convs that never take it by 15-25 % (A/B, same box)
        // ConvTranspose1d(k == stride == up): the 4 consecutive rows (co, j) a lane holds for its column t are up consecutive
        // samples of 4 / up channels -- one 8- or 16-B store per channel, 16 lanes = one contiguous run, instead of 4-B
        // stores scattered at a stride of up floats
        const size_t Lout = (size_t)p.Tout * p.up;
        const int Cq = p.Co / p.up;
#pragma unroll
        for (int nn = 0; nn < NR; ++nn) {
            const int t = t0 + wn * TN + nn * 16 + lr;
            if (t >= p.Tout) continue;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const int row = m0 + wm * TM + m * 16 + lq * 4;          // multiple of 4, Co is a multiple of up
                if (p.up == 4) {
                    if (row < p.Co) *(f32x4*)(p.Y + ((size_t)n * Cq + (row >> 2)) * Lout + (size_t)t * 4) = acc[m][nn];
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (row + 2 * h < p.Co) {
                            f32x2 v = {acc[m][nn][2 * h], acc[m][nn][2 * h + 1]};
                            *(f32x2*)(p.Y + ((size_t)n * Cq + (row >> 1) + h) * Lout + (size_t)t * 2) = v;
                        }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int nn = 0; nn < NR; ++nn) {
        const int t = t0 + wn * TN + nn * 16 + lr;
        if (t >= p.Tout) continue;
        Lerp lp;
        if (p.Z != nullptr) lp = lerp_coord(t, film_ratio, p.Lf);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row_local = wm * TM + m * 16 + lq * 4 + r;
                if (m0 + row_local >= p.Co) continue;
                conv_epilogue_store<false>(p, n, m0 + row_local, row_local, t, acc[m][nn][r], lp, ft);
            }
        }
        if constexpr (YPL) {
            // Planes of this column: a lane holds, per 16-row group m, four consecutive channels (lq * 4 ..) -- 8 bytes per plane.  The
            // groups m, m + 1 of a 32-channel k-block are exchanged between the lanes lq, lq ^ 1 (same column: lane ^ 16) so that an even
            // lq owns channels lq * 4 .. + 7 and an odd one 16 + (lq - 1) * 4 .. + 7: one 16-byte store per lane, plane and k-block, and a
            // wave's store instruction writes 16 columns x 64 B = one contiguous 1-KB run (8-byte stores touched 32 B of every line twice:
            // 1.93 -> 3.25 ms per step for this conv).  hi = bf16(v), lo = bf16(v - hi): alive_to_planes's split.
            static_assert(!YPL || (MR % 2) == 0, "the plane image pairs the 16-row groups");
            const int c_pad = (p.Co + 31) & ~31;
            const int64_t cols_pad = (((int64_t)p.N * p.Tout + 127) / 128) * 128;
            unsigned short* Po = (unsigned short*)p.Yp;
            const bool odd = (lq & 1) != 0;
#pragma unroll
            for (int mp = 0; mp < MR; mp += 2) {
                const int rbase = m0 + wm * TM + mp * 16;                 // first channel of the k-block (wave-uniform)
                if (rbase >= c_pad) continue;                              // (channels Co .. c_pad - 1: zeros, as alive_to_planes leaves them)
                float qa[4], qb[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qa[r] = rbase + lq * 4 < p.Co ? acc[mp][nn][r] : 0.0f;
                    qb[r] = rbase + 16 + lq * 4 < p.Co ? acc[mp + 1][nn][r] : 0.0f;
                }
                const bool one_f16 = p.yp_planes == 1;                      // ONE fp16 plane (the plain consumer) instead of two bf16 planes
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    if (pl == 1 && one_f16) break;
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    auto pk = [one_f16](float a, float b) {
                        const bf16x2_t h = {(__bf16)a, (__bf16)b};
                        return one_f16 ? pack_f16x2(a, b) : __builtin_bit_cast(unsigned, h);
                    };
                    const unsigned a01 = pk(qa[0], qa[1]), a23 = pk(qa[2], qa[3]), b01 = pk(qb[0], qb[1]), b23 = pk(qb[2], qb[3]);
                    const unsigned rx = (unsigned)__shfl_xor((int)(odd ? a01 : b01), 16), ry = (unsigned)__shfl_xor((int)(odd ? a23 : b23), 16);
                    const u32x4 o = odd ? u32x4{rx, ry, b01, b23} : u32x4{a01, a23, rx, ry};
                    const int ch = rbase + (odd ? 16 + (lq - 1) * 4 : lq * 4);
                    *(u32x4*)(Po + planes_at(pl, (int64_t)n * p.Tout + t, ch, cols_pad, c_pad)) = o;
                    qa[0] -= __uint_as_float(a01 << 16); qa[1] -= __uint_as_float(a01 & 0xffff0000u);
                    qa[2] -= __uint_as_float(a23 << 16); qa[3] -= __uint_as_float(a23 & 0xffff0000u);
                    qb[0] -= __uint_as_float(b01 << 16); qb[1] -= __uint_as_float(b01 & 0xffff0000u);
                    qb[2] -= __uint_as_float(b23 << 16); qb[3] -= __uint_as_float(b23 & 0xffff0000u);
                }
            }
        }
    }
    if constexpr (YPL) {
        // the padding columns N Tout .. cols_pad - 1 (fewer than 128): zeros, written by the block of the last column tile
        if (n == p.N - 1 && blockIdx.x == gridDim.x - 1) {
            const int c_pad = (p.Co + 31) & ~31;
            const int64_t cols = (int64_t)p.N * p.Tout, cols_pad = ((cols + 127) / 128) * 128;
            const int per_col = (p.yp_planes == 1 ? 1 : 2) * (c_pad / 4);      // 8-byte pieces per column: planes x channel quads
            unsigned short* Po = (unsigned short*)p.Yp;
            for (int i = threadIdx.x; i < (int)(cols_pad - cols) * per_col; i += 256) {
                const int c = i / per_col, w = i % per_col;
                *(uint2*)(Po + planes_at(w / (c_pad / 4), cols + c, (w % (c_pad / 4)) * 4, cols_pad, c_pad)) = make_uint2(0u, 0u);
            }
        }
    }
}

}  // namespace

int alive_conv_split_launch(const AliveConv* d, float ratio, hipStream_t s);
bool alive_conv_skinny_try(const AliveConv* d, hipStream_t s, int* rc);

extern "C" int alive_conv1d(const AliveConv* d, void* stream) {
    ALIVE_CHECK_ARG(d && d->W && (d->X || d->Xp), "alive_conv1d: null W/X");
    ALIVE_CHECK_ARG(d->N > 0 && d->Ci > 0 && d->Co > 0 && d->Tin > 0 && d->Tout > 0, "alive_conv1d: bad sizes");
    ALIVE_CHECK_ARG(d->KW >= 1 && (d->KW <= 16 || d->Ci == 1) && d->stride >= 1 && d->dil >= 1 && d->up >= 1, "alive_conv1d: bad geometry");
    ALIVE_CHECK_ARG(d->precision >= 1 || (d->K_pad % 16 == 0 && d->K_pad >= d->Ci * d->KW), "alive_conv1d: K_pad %d for K %d", d->K_pad, d->Ci * d->KW);
    ALIVE_CHECK_ARG(d->precision >= 0 && d->precision <= 3, "alive_conv1d: precision");
    ALIVE_CHECK_ARG(d->Ci * d->KW < 32768, "alive_conv1d: K too large");
    ALIVE_CHECK_ARG(d->Y || d->Z || d->Zp, "alive_conv1d: no output");
    if (d->up > 1) {
        ALIVE_CHECK_ARG(d->Co % d->up == 0 && !d->residual && !d->skip && !d->Z && !d->Zp && d->Y, "alive_conv1d: transposed conv has a plain epilogue");
    }
    ALIVE_CHECK_ARG(d->yp_planes >= 0 && d->yp_planes <= 2, "alive_conv1d: yp_planes");
    if (d->Yp)
        ALIVE_CHECK_ARG(d->precision == 0 && d->Y && d->up == 1 && d->act == 0 && !d->post_add && !d->ch_scale && !d->residual && !d->skip && !d->Z &&
                        !d->Zp && d->Co > 16 && d->Co <= 64 && (d->Co & 3) == 0 && (((uintptr_t)d->Yp) & 15) == 0 && (int64_t)d->N * d->Tout >= 97,
                        "alive_conv1d: Yp (plane image of Y) needs the exact kernel's plain conv with 16 < Co <= 64, Co %% 4 == 0, more than 96 columns");
    ALIVE_CHECK_ARG(d->pad_mode >= 0 && d->pad_mode <= 2, "alive_conv1d: pad_mode");
    if (d->pad_mode != 0) ALIVE_CHECK_ARG(d->pad_left < d->Tin, "alive_conv1d: reflect pad %d needs Tin > pad (Tin %d)", d->pad_left, d->Tin);
    if (d->Z || d->Zp) ALIVE_CHECK_ARG(d->film && d->Lf > 0, "alive_conv1d: Z needs film");
    if (d->precision == 3)
        ALIVE_CHECK_ARG((int64_t)d->N * d->Tout > 96, "alive_conv1d: precision 3 (plain fp16) is a batch form: more than 96 columns, got %lld",
                        (long long)d->N * d->Tout);
    if (d->Xp || d->Zp) ALIVE_CHECK_ARG(d->precision == 1 || d->precision == 3, "alive_conv1d: plane-packed operands need the split kernel (precision 1 or 3)");
    {
        int rc;
        if (d->film_ld == 0 && !d->Xp && !d->Zp && alive_conv_skinny_try(d, (hipStream_t)stream, &rc)) return rc;      // few columns (streaming)
    }
    if (d->precision >= 1)
        return alive_conv_split_launch(d, (d->Z || d->Zp) ? (float)(d->film_ld ? d->film_ld : d->Lf) / (float)d->Tout : 0.0f, (hipStream_t)stream);
    ALIVE_CHECK_ARG(d->film_ld == 0, "alive_conv1d: a film frame range needs the split kernel (precision 1 / 2)");
    const unsigned magic = d->Ci == 1 ? 0u : (unsigned)(((1u << 20) + d->KW - 1) / d->KW);
    const float ratio = d->Z ? (float)d->Lf / (float)d->Tout : 0.0f;
    hipStream_t s = (hipStream_t)stream;
    if (d->Co > 64) {
        dim3 g(cdiv(d->Tout, 128), cdiv(d->Co, 128), d->N);
        conv_gemm_kernel<128, 128, 2, 2><<<g, 256, 0, s>>>(*d, magic, ratio);
    } else if (d->Co > 16) {
        dim3 g(cdiv(d->Tout, 128), cdiv(d->Co, 64), d->N);
        if (d->Yp) conv_gemm_kernel<64, 128, 1, 4, false, true><<<g, 256, 0, s>>>(*d, magic, ratio);
        else conv_gemm_kernel<64, 128, 1, 4><<<g, 256, 0, s>>>(*d, magic, ratio);
    } else {
        dim3 g(cdiv(d->Tout, 256), 1, d->N);
        if ((d->up == 2 || d->up == 4) && d->act == 0 && !d->post_add && !d->ch_scale)
            conv_gemm_kernel<16, 256, 1, 4, true><<<g, 256, 0, s>>>(*d, magic, ratio);      // vector stores of the transposed conv
        else
            conv_gemm_kernel<16, 256, 1, 4><<<g, 256, 0, s>>>(*d, magic, ratio);
    }
    ALIVE_CHECK_LAUNCH("alive_conv1d");
    return ALIVE_OK;
}

ALIVE_F16_SAT_GETTER(alive_f16_sat_conv)
