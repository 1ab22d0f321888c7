// The pointwise half of a ConvNeXt layer of the decoder's FeatureExtractor -- x <- x + scale * pw2(gelu(pw1(y))) (/root/reference/module/
// common.py:74-82; y = the normalised depthwise output, which alive_dwconv_norm_planes leaves as ONE fp16 plane) -- in ONE kernel on
// plain fp16 operands (decoder precision mode 1, batch path): round 6.
//
// As two plane GEMMs (gemm_planes_kernel<.., 1, ..>: 8.2 ms per step for the four layers) the hidden layer -- 1536 channels, three
// times the input -- is written as an fp16 plane and read back: 6 of the 8 KB a frame moves per layer, for GEMMs that then sit at 0.3 of
// the matrix pipe and 1.9 TB/s.  Here a block keeps a tile of 128 frames on chip:
//   * y[512 channels][128 frames] as one fp16 plane in LDS (128 KB: [frame][512 channels] rows of 1 KB, 16-byte chunks XOR-swizzled by
//     (row & 15): a 32x32x16 B fragment is one conflict-free ds_read_b128);
//   * the hidden layer in chunks of 128 channels: chunk c = gelu(W1[c] y + b1[c]) -- wave w the 32 rows 32 w .., K = 512 -- goes to a
//     32-KB LDS buffer as fp16 ([frame][128 channels]), and at once feeds Y += W2[:, c] chunk (wave w the 128 output rows 128 w .., K =
//     128): the hidden layer never leaves the chip;
//   * the output tile's 512 x 128 fp32 accumulators stay in registers through the twelve chunks (256 per lane), then (Y + b2) * scale +
//     x goes back to x;
//   * W1 and W2 stream from L2 as fragment-shaped reads of the k-blocked fp16 slabs of module/_pack.py::pack_conv_split_h, every
//     fragment once per block -- 3 MB per 128 frames, the ratio of the fused FilterBlock (filter_big.hip); one memory instruction
//     behind each MFMA or two.
// Same products in the same order as the two GEMMs, the same bias / GELU / conversion sequence (gemm_planes.hip::gemm_epilogue,
// conv_epilogue.h::gelu_fast2): the result is theirs bit for bit (tools/cmp_mlp.py).  ALIVE_MLP_FUSED=0 keeps the two GEMMs.
#include "conv_epilogue.h"
#include "lds_as3.h"
#include <stdlib.h>

namespace {

constexpr int CT = 3;                     // column tiles of 32 frames per tile: all MFMA accumulators of a 512-register kernel must be AGPRs
                                          // (hipcc picks one MFMA form per function), and 4 x 4 output + 4 hidden tiles are 320 of the 256
constexpr int BL = 32 * CT;               // frames per tile
constexpr int HC = 128;                   // hidden channels per chunk
constexpr int PF1 = 4, PF2 = 4;           // k-steps of weights in flight (first / second GEMM)

template <int C, int H>
struct MlpGeo {
    static constexpr int XROW = 2 * C;                         // 1 KB: a frame's C channels, fp16
    static constexpr int HROW = 2 * HC;                        // 256 B
    static constexpr int XBYTES = BL * XROW, HBYTES = BL * HROW;
    static constexpr int LDS_BYTES = XBYTES + HBYTES;          // 160 KB at C = 512
    static constexpr int NCH = H / HC;                         // 12 chunks
    static constexpr int KS1 = C / 16, KS2 = HC / 16;          // 32 / 8 k-steps
    static_assert(C == 128 * 4 && H % HC == 0 && LDS_BYTES <= 160 * 1024 && CT <= 4, "geometry");
};

template <int C, int H>
__global__ __launch_bounds__(256, 1) void convnext_mlp_kernel(const unsigned short* __restrict__ P, int64_t cols, int64_t cols_pad, int T,
                                                              const unsigned short* __restrict__ W1, const float* __restrict__ b1,
                                                              const unsigned short* __restrict__ W2, const float* __restrict__ b2,
                                                              const float* __restrict__ ch_scale, float* __restrict__ x) {
    typedef MlpGeo<C, H> G;
    constexpr int XROW = G::XROW, HROW = G::HROW, KS1 = G::KS1, KS2 = G::KS2, NCH = G::NCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int sm0 = (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sm;
    const int Xs = sm0, Hs = sm0 + G::XBYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, lh = lane >> 5;
    const int64_t c0 = (int64_t)blockIdx.x * BL;

    // ---- the tile of y: 16 k-blocks x (BL frames x 64 B, contiguous in the k-blocked plane) -> LDS ----
    constexpr int PPK = BL * 4;                       // 16-byte pieces per k-block
#pragma unroll
    for (int q4 = 0; q4 < C / 32 / 4; ++q4) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kb = 4 * q4 + (j >> 1), piece = tid + 256 * (j & 1);
            const int col = piece >> 2, sub = piece & 3;
            int64_t gc = c0 + col;
            gc = gc < cols_pad ? gc : cols_pad - 1;
            v[j] = piece < PPK ? __builtin_nontemporal_load((const u32x4*)(P + (((size_t)kb * cols_pad + gc) * 32 + sub * 8))) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kb = 4 * q4 + (j >> 1), piece = tid + 256 * (j & 1);
            const int col = piece >> 2, sub = piece & 3;
            if (piece < PPK) lds_put<u32x4>(Xs + col * XROW + (((4 * kb + sub) ^ (col & 15)) << 4), 0, v[j]);
        }
    }

    // weights: A fragments.  First GEMM: rows c HC + 32 w + n32 of W1 [C / 32][H][32]; second: rows 128 w + 32 rg + n32 of W2 [H / 32][C][32]
    const unsigned short* w1row = W1 + (size_t)(32 * w + n32) * 32 + 8 * lh;
    const unsigned short* w2row = W2 + (size_t)(128 * w + n32) * 32 + 8 * lh;
    auto a1_ptr = [&](int c, int ks) { return (const bf16x8*)(w1row + ((size_t)(ks >> 1) * H + c * HC) * 32 + (ks & 1) * 16); };
    auto a2_ptr = [&](int c, int ks, int rg) { return (const bf16x8*)(w2row + ((size_t)(c * (HC / 32) + (ks >> 1)) * C + 32 * rg) * 32 + (ks & 1) * 16); };
    // B fragments: column tile ct of frame rows 32 ct + n32 (the swizzle term is the same for all four: 32 rows on)
    const int xb = Xs + n32 * XROW, hb = Hs + n32 * HROW, sw = n32 & 15;
    auto bx = [&](int ks, int ct) { return lds_get<bf16x8>(xb + (((2 * ks + lh) ^ sw) << 4), 32 * XROW * ct); };
    auto bh = [&](int ks, int ct) { return lds_get<bf16x8>(hb + (((2 * ks + lh) ^ sw) << 4), 32 * HROW * ct); };

    f32x16 Y[4][CT];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[rg][ct][r] = 0.0f;

    bf16x8 a1[PF1];
#pragma unroll
    for (int s = 0; s < PF1; ++s) a1[s] = *a1_ptr(0, s);
    __syncthreads();                                  // the tile of y is in LDS

#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        // ---- hidden chunk c: 32 rows x 128 frames per wave, K = C ----
        f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
        bf16x8 bfr[2][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) bfr[0][ct] = bx(0, ct);
#pragma unroll 4
        for (int ks = 0; ks < KS1; ++ks) {
            const int k1 = ks + 1 < KS1 ? ks + 1 : KS1 - 1;
            const int kn = ks + PF1 < KS1 ? ks + PF1 : KS1 - 1;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                acc[ct] = mfma_f16(a1[ks % PF1], bfr[ks & 1][ct], acc[ct]);
                __builtin_amdgcn_sched_barrier(0);
                bfr[(ks + 1) & 1][ct] = bx(k1, ct);
                __builtin_amdgcn_sched_barrier(0);
            }
            a1[ks % PF1] = *a1_ptr(c, kn);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the second GEMM's first fragments, and the next chunk's, in flight under the epilogue
        bf16x8 a2[PF2][4];
#pragma unroll
        for (int s = 0; s < PF2; ++s)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) a2[s][rg] = *a2_ptr(c, s, rg);
        if (c + 1 < NCH) {
#pragma unroll
            for (int s = 0; s < PF1; ++s) a1[s] = *a1_ptr(c + 1, s);
        }
        // ---- + bias -> gelu -> fp16 -> the chunk buffer (gemm_epilogue's sequence for a plane output) ----
        {
            f32x4 bia[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bia[g] = *(const f32x4*)(b1 + c * HC + 32 * w + 8 * g + 4 * lh);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int col = 32 * ct + n32;
                const bool cok = c0 + col < cols;              // (fp16 saturations of real frames only)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x2 g01 = gelu_fast2(f32x2{acc[ct][4 * g] + bia[g][0], acc[ct][4 * g + 1] + bia[g][1]});
                    const f32x2 g23 = gelu_fast2(f32x2{acc[ct][4 * g + 2] + bia[g][2], acc[ct][4 * g + 3] + bia[g][3]});
                    const u32x2 hh = {pack_f16x2((g01[0] + 0.0f) * 1.0f, (g01[1] + 0.0f) * 1.0f, cok), pack_f16x2((g23[0] + 0.0f) * 1.0f, (g23[1] + 0.0f) * 1.0f, cok)};
                    lds_put<u32x2>(Hs + col * HROW + (((4 * w + g) ^ (col & 15)) << 4) + 8 * lh, 0, hh);
                }
            }
        }
        __syncthreads();                              // the chunk is complete
        // ---- Y += W2[:, chunk] x chunk: 128 rows x 128 frames per wave, K = 128 ----
        bf16x8 hfr[2][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) hfr[0][ct] = bh(0, ct);
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            const int k1 = ks + 1 < KS2 ? ks + 1 : KS2 - 1;
            const int kn = ks + PF2 < KS2 ? ks + PF2 : KS2 - 1;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    Y[rg][ct] = mfma_f16(a2[ks % PF2][rg], hfr[ks & 1][ct], Y[rg][ct]);
                    if (ct == 1 && rg < CT) {              // (a memory instruction per two MFMAs: the next k-step's B fragments, this row group's refill)
                        __builtin_amdgcn_sched_barrier(0);
                        hfr[(ks + 1) & 1][rg] = bh(k1, rg);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                a2[ks % PF2][rg] = *a2_ptr(c, kn, rg);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                              // every wave is done reading the chunk buffer
    }

    // ---- x <- (Y + b2) * scale + x   (gemm_epilogue's order of operations) ----
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int64_t col = c0 + 32 * ct + n32;
        if (col >= cols) continue;
        const int n = (int)(col / T), t = (int)(col - (int64_t)n * T);
        unsigned lo = (unsigned)(4 * lh) * (unsigned)T + (unsigned)t;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            float* xb_ = x + ((size_t)n * C + 128 * w + 32 * rg) * T;
            float res[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = __builtin_nontemporal_load(xb_ + (size_t)(8 * (r >> 2) + (r & 3)) * T + lo);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 128 * w + 32 * rg + 8 * (r >> 2) + 4 * lh + (r & 3);
                float v = Y[rg][ct][r] + b2[row];
                v = (v + 0.0f) * ch_scale[row];
                v += res[r];
                __builtin_nontemporal_store(v, xb_ + (size_t)(8 * (r >> 2) + (r & 3)) * T + lo);
            }
        }
    }
}

}  // namespace

// P: ONE fp16 plane of the normalised activations, k-blocked [C / 32][cols_pad][32] (alive_dwconv_norm_planes, planes = 1), frames
// n * T + t, cols_pad = N * T rounded up to 128;  W1 / W2: the fp16 slabs of module/_pack.py::pack_conv_split_h ([C / 32][H][32],
// [H / 32][C][32]);  x[N][C][T] fp32: the layer's input and output (residual).  C = 512, H = 1536.
extern "C" int alive_convnext_mlp_fp16(const void* P, int N, int T, int C, int H, const void* W1, const float* b1, const void* W2, const float* b2,
                                       const float* ch_scale, float* x, void* stream) {
    ALIVE_CHECK_ARG(P && W1 && b1 && W2 && b2 && ch_scale && x, "alive_convnext_mlp_fp16: null pointer");
    ALIVE_CHECK_ARG(N > 0 && T > 0, "alive_convnext_mlp_fp16: bad sizes");
    ALIVE_CHECK_ARG(C == 512 && H == 1536, "alive_convnext_mlp_fp16: built for 512 -> 1536 -> 512 (got %d -> %d)", C, H);
    typedef MlpGeo<512, 1536> G;
    {
        static LdsOptIn optin;
        hipError_t e = optin.ensure({(const void*)convnext_mlp_kernel<512, 1536>}, G::LDS_BYTES);
        if (e != hipSuccess) {
            alive_set_error("alive_convnext_mlp_fp16: cannot reserve %d B of LDS: %s", G::LDS_BYTES, hipGetErrorString(e));
            return ALIVE_ERR_LAUNCH;
        }
    }
    const int64_t cols = (int64_t)N * T, cols_pad = (cols + 127) / 128 * 128;
    ALIVE_CHECK_ARG(cols / BL + 1 < (1ll << 31), "alive_convnext_mlp_fp16: too many frames");
    convnext_mlp_kernel<512, 1536><<<(unsigned)((cols + BL - 1) / BL), 256, G::LDS_BYTES, (hipStream_t)stream>>>(
        (const unsigned short*)P, cols, cols_pad, T, (const unsigned short*)W1, b1, (const unsigned short*)W2, b2, ch_scale, x);
    ALIVE_CHECK_LAUNCH("alive_convnext_mlp_fp16");
    return ALIVE_OK;
}

ALIVE_F16_SAT_GETTER(alive_f16_sat_mlp)
