// Convs over very few GEMM columns (N*T <= 96: the streaming path of realtime_inference.py, 5-24 frames per step).
//
// At 8 frames a 512x1536 GEMM has 12 M MACs and 3 MB of weights: the tiled kernels run it as a handful of blocks that
// walk K serially behind LDS round trips and barriers (70-95 us per launch, 33 launches per step).  Here the weight
// matrix is cut into 16-row slabs, one block each (32..258 blocks), the block's eight waves split K between them and
// stream their weights straight from L2 into f32-MFMA A fragments (v_mfma_f32_16x16x4_f32), the partial tiles are
// summed through LDS and go through the common epilogue.  Weights are read in whatever format the layer was packed
// for the tiled kernels: fp32 [Co_pad][K_pad], or 2 / 3 bf16 planes whose sum is the (16- / 24-bit mantissa) weight.
#include "conv_epilogue.h"
#include "planes_layout.h"
#include <stdlib.h>

namespace {

// x value of GEMM column (n, t) at reduction index (ci, tap j): the implicit im2col of alive_conv1d.  Branch-free (one
// unconditional load from a clamped address) so that the loads of several k-steps can be in flight together.
__device__ __forceinline__ float skinny_x(const AliveConv& p, const float* xn, int ci, int t, int j) {
    int tin = t * p.stride + j * p.dil - p.pad_left;
    bool ok = ci < p.Ci;
    ok = ok && !(tin < 0 && p.pad_mode == 0);
    tin = tin < 0 ? -tin : tin;                            // reflect (left: modes 1 and 2)
    ok = ok && !(tin >= p.Tin && p.pad_mode != 2);
    tin = tin >= p.Tin ? 2 * (p.Tin - 1) - tin : tin;      // reflect right (STFT centre pad)
    ok = ok && tin >= 0 && tin < p.Tin;
    const float v = xn[ok ? (size_t)ci * p.Tin + tin : 0];
    return ok ? v : 0.0f;
}

// waves per block = K-split: 8, or 16 where a wave would otherwise walk 4 or more dependent k-steps (K >= 512: the 1x1 convs
// of the ConvNeXt layers, the k5 convs of the 256-channel FilterBlock, the strided down convs) -- these launches are latency chains of
// load -> MFMA on 16 .. 48 CUs, and the chain is what a streaming step waits for

// PW: a pointwise conv (KW == 1, stride 1, no padding: two thirds of a streaming step's launches) -- the im2col index is ci * Tin + t,
// not the ~15 vector instructions of skinny_x per loaded value (round 4: they, not the loads or the MFMAs, were most of the kernel)
template <int NP, bool PW, int SKW>
__global__ __launch_bounds__(64 * SKW) void conv_skinny_kernel(AliveConv p, int ncols) {
    __shared__ f32x4 red[2][SKW][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ln = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int c0 = blockIdx.y * 32;                    // first GEMM column of this block
    const int co_pad = (p.Co + 15) & ~15;
    const int Kp = NP == 0 ? p.K_pad : p.KW * p.Ci_pad;    // padded K of the weight rows (zero padded)
    int row = m0 + ln;
    row = row < co_pad ? row : co_pad - 1;

    // columns: c = n * Tout + t  (two 16-column MFMA tiles)
    const float* xn[2];
    int tcol[2];
    bool cok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int c = c0 + h * 16 + ln;
        cok[h] = c < ncols;
        int n = cok[h] ? c / p.Tout : 0;
        tcol[h] = cok[h] ? c - n * p.Tout : 0;
        xn[h] = p.X + (size_t)n * p.Ci * p.Tin;
    }
    const bool two = c0 + 16 < ncols;
    // wave 0 starts from the bias (a K == 1 conv is then the single fma(w, x, b) of the tiled kernel: F0Encoder.c1)
    f32x4 acc[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = m0 + lq * 4 + r;
        const float b = (wv == 0 && p.bias != nullptr && orow < p.Co) ? p.bias[orow] : 0.0f;
        acc[0][r] = b;
        acc[1][r] = b;
    }
    // wave wv walks k = 16*(wv + SKW*i) .. +15 ; lane (row ln, slot lq) holds k0 + 4*lq + s for MFMA step s.
    // fp32 weights: k = ci * KW + j;  bf16 planes: tap-major k = j * Ci_pad + ci (a 16-k group lies inside one tap)
    for (int k0 = wv * 16; k0 < Kp; k0 += 16 * SKW) {
        f32x4 w;
        if (NP == 0) {
            w = *(const f32x4*)(p.W + (size_t)row * p.K_pad + k0 + 4 * lq);
        } else {
            const unsigned short* W16 = (const unsigned short*)p.W;
            w = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int pl = NP - 1; pl >= 0; --pl) {      // smallest plane first: the partial sums stay exact
                const uint2 q = *(const uint2*)(W16 + planes_at(pl, row, k0 + 4 * lq, co_pad, Kp));      // k-blocked planes
                w[0] += __uint_as_float(q.x << 16);
                w[1] += __uint_as_float(q.x & 0xffff0000u);
                w[2] += __uint_as_float(q.y << 16);
                w[3] += __uint_as_float(q.y & 0xffff0000u);
            }
        }
        float x[2][4];
        const int jt = NP == 0 ? 0 : k0 / p.Ci_pad, cbase = NP == 0 ? 0 : k0 - jt * p.Ci_pad;
        // plane-packed weights are tap-major: the 16 k of a step share one tap, so the padding / reflection logic of skinny_x runs
        // once per column here instead of once per loaded value
        int tin_h[2] = {0, 0};
        bool ok_h[2] = {false, false};
        if (NP != 0 && !PW) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int tin = tcol[h] * p.stride + jt * p.dil - p.pad_left;
                bool ok = cok[h] && !(tin < 0 && p.pad_mode == 0);
                tin = tin < 0 ? -tin : tin;
                ok = ok && !(tin >= p.Tin && p.pad_mode != 2);
                tin = tin >= p.Tin ? 2 * (p.Tin - 1) - tin : tin;
                ok_h[h] = ok && tin >= 0 && tin < p.Tin;
                tin_h[h] = tin;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + 4 * lq + s;
            int ci, j;
            if (NP == 0) {
                ci = p.KW == 1 ? k : k / p.KW;
                j = k - ci * p.KW;
            } else {
                ci = cbase + 4 * lq + s;
                j = jt;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (PW) {
                    const bool ok = cok[h] && ci < p.Ci;
                    const float v = xn[h][ok ? (size_t)ci * p.Tin + tcol[h] : 0];
                    x[h][s] = ok ? v : 0.0f;
                } else if (NP != 0) {
                    const bool ok = ok_h[h] && ci < p.Ci;
                    const float v = xn[h][ok ? (size_t)ci * p.Tin + tin_h[h] : 0];
                    x[h][s] = ok ? v : 0.0f;
                } else {
                    x[h][s] = cok[h] ? skinny_x(p, xn[h], ci, tcol[h], j) : 0.0f;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], x[0][s], acc[0], 0, 0, 0);
            if (two) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], x[1][s], acc[1], 0, 0, 0);
        }
    }
    red[0][wv][lane] = acc[0];
    red[1][wv][lane] = acc[1];
    __syncthreads();
    // wave h (0, 1) finishes column tile h: sum of the K-parts + bias, then the common epilogue
    if (wv < 2 && c0 + wv * 16 < ncols) {
        f32x4 v = red[wv][0][lane];
#pragma unroll
        for (int i = 1; i < SKW; ++i) v = v + red[wv][i][lane];
        const int c = c0 + wv * 16 + ln;
        if (c < ncols) {
            const int n = c / p.Tout, t = c - n * p.Tout;
            Lerp lp;
            if (p.Z != nullptr) lp = lerp_coord(t, (float)p.Lf / (float)p.Tout, p.Lf);
            FilmTile ft{nullptr, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = m0 + lq * 4 + r;
                if (orow < p.Co) {
                    conv_epilogue_store<false>(p, n, orow, lq * 4 + r, t, v[r], lp, ft);
                }
            }
        }
    }
}

}  // namespace

// true if the descriptor qualifies; launches the skinny kernel.  Any conv of alive_conv1d over at most SKINNY_COLS GEMM
// columns: the 1x1 convs, the DFT, the strided down-convs, the k5 filter convs and the transposed convs of a
// streaming step (8 frames: 8 .. 80 columns below the 16-channel scale).
constexpr int SKINNY_COLS = 96;
bool alive_conv_skinny_try(const AliveConv* d, hipStream_t s, int* rc) {
    const int ncols = d->N * d->Tout;
    if (ncols > SKINNY_COLS || d->precision == 3) return false;      // (plain fp16: batch kernels only, alive_conv1d checks the columns)
    if (d->precision == 0 && (d->K_pad & 15)) return false;
    dim3 g(cdiv(d->Co, 16), cdiv(ncols, 32));
    const bool pw = d->KW == 1 && d->stride == 1 && d->pad_left == 0 && d->Tout <= d->Tin;
    static const int k16 = getenv("ALIVE_SKINNY_K16") ? atoi(getenv("ALIVE_SKINNY_K16")) : 512;
    const int Kp = d->precision == 0 ? d->K_pad : d->KW * d->Ci_pad;
    const bool wide = Kp >= k16;
#define SKINNY_LAUNCH(NP_, PW_) do { if (wide) conv_skinny_kernel<NP_, PW_, 16><<<g, 1024, 0, s>>>(*d, ncols); \
                                      else conv_skinny_kernel<NP_, PW_, 8><<<g, 512, 0, s>>>(*d, ncols); } while (0)
    if (d->precision == 0) { if (pw) SKINNY_LAUNCH(0, true); else SKINNY_LAUNCH(0, false); }
    else if (d->precision == 1) { if (pw) SKINNY_LAUNCH(2, true); else SKINNY_LAUNCH(2, false); }
    else { if (pw) SKINNY_LAUNCH(3, true); else SKINNY_LAUNCH(3, false); }
#undef SKINNY_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        alive_set_error("alive_conv1d(skinny): %s", hipGetErrorString(e));
        *rc = ALIVE_ERR_LAUNCH;
    } else {
        *rc = ALIVE_OK;
    }
    return true;
}
