// 1x1 convs over very few columns (N*T <= 32: the streaming path of realtime_inference.py, 5-24 frames per step).
//
// At 8 frames a 512x1536 GEMM has 12 M MACs and 3 MB of weights: the tiled kernels run it as a handful of blocks that
// walk K serially behind LDS round trips and barriers (70-95 us per launch, 33 launches per step).  Here the weight
// matrix is cut into 16-row slabs, one block each (32..258 blocks), the block's four waves split K between them and
// stream their weights straight from L2 into f32-MFMA A fragments (v_mfma_f32_16x16x4_f32), the four partial tiles are
// summed through LDS and go through the common epilogue.  Weights are read in whatever format the layer was packed
// for the tiled kernels: fp32 [Co_pad][K_pad], or 2 / 3 bf16 planes whose sum is the (16- / 24-bit mantissa) weight.
#include "conv_epilogue.h"

namespace {

template <int NP>       // 0: fp32 weights; 2 / 3: bf16 planes
__device__ __forceinline__ f32x4 load_w4(const AliveConv& p, int row, int k, int co_pad) {
    if (NP == 0) {
        return *(const f32x4*)(p.W + (size_t)row * p.K_pad + k);
    } else {
        const unsigned short* W16 = (const unsigned short*)p.W;
        f32x4 w = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int pl = NP - 1; pl >= 0; --pl) {          // smallest plane first: the partial sums stay exact
            const uint2 q = *(const uint2*)(W16 + ((size_t)pl * co_pad + row) * p.Ci_pad + k);
            w[0] += __uint_as_float(q.x << 16);
            w[1] += __uint_as_float(q.x & 0xffff0000u);
            w[2] += __uint_as_float(q.y << 16);
            w[3] += __uint_as_float(q.y & 0xffff0000u);
        }
        return w;
    }
}

template <int NP>
__global__ __launch_bounds__(256) void conv_skinny_kernel(AliveConv p, int ncols) {
    __shared__ f32x4 red[2][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ln = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int co_pad = (p.Co + 15) & ~15;
    const int Kp = NP == 0 ? p.K_pad : p.Ci_pad;       // padded K of the weight rows (zero padded)
    int row = m0 + ln;
    row = row < co_pad ? row : co_pad - 1;

    // columns: c = n * Tout + t  (two 16-column MFMA tiles)
    const float* xcol[2];
    bool cok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int c = h * 16 + ln;
        cok[h] = c < ncols;
        int n = cok[h] ? c / p.Tout : 0, t = cok[h] ? c - n * p.Tout : 0;
        xcol[h] = p.X + (size_t)n * p.Ci * p.Tin + t;
    }
    // wave 0 starts from the bias (a K == 1 conv is then the single fma(w, x, b) of the tiled kernel: F0Encoder.c1)
    f32x4 acc[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = m0 + lq * 4 + r;
        const float b = (wv == 0 && p.bias != nullptr && orow < p.Co) ? p.bias[orow] : 0.0f;
        acc[0][r] = b;
        acc[1][r] = b;
    }
    // wave wv walks k = 16*(wv + 4*i) .. +15 ; lane (row ln, slot lq) holds k0 + 4*lq + s for MFMA step s
    for (int k0 = wv * 16; k0 < Kp; k0 += 64) {
        const f32x4 w = load_w4<NP>(p, row, k0 + 4 * lq, co_pad);
        float x[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = k0 + 4 * lq + s;
                x[h][s] = (cok[h] && k < p.Ci) ? xcol[h][(size_t)k * p.Tin] : 0.0f;
            }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], x[0][s], acc[0], 0, 0, 0);
            if (ncols > 16) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], x[1][s], acc[1], 0, 0, 0);
        }
    }
    red[0][wv][lane] = acc[0];
    red[1][wv][lane] = acc[1];
    __syncthreads();
    // wave h (0, 1) finishes column tile h: sum of the four K-quarters + bias, then the common epilogue
    if (wv < 2 && wv * 16 < ncols) {
        f32x4 v = red[wv][0][lane] + red[wv][1][lane] + red[wv][2][lane] + red[wv][3][lane];
        const int c = wv * 16 + ln;
        if (c < ncols) {
            const int n = c / p.Tout, t = c - n * p.Tout;
            Lerp lp;
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

// true if the descriptor qualifies; launches the skinny kernel
bool alive_conv_skinny_try(const AliveConv* d, hipStream_t s, int* rc) {
    const int ncols = d->N * d->Tout;
    if (d->KW != 1 || d->stride != 1 || ncols > 32 || d->Z != nullptr || d->Tout != d->Tin) return false;
    dim3 g(cdiv(d->Co, 16));
    if (d->precision == 0) conv_skinny_kernel<0><<<g, 256, 0, s>>>(*d, ncols);
    else if (d->precision == 1) conv_skinny_kernel<2><<<g, 256, 0, s>>>(*d, ncols);
    else conv_skinny_kernel<3><<<g, 256, 0, s>>>(*d, ncols);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        alive_set_error("alive_conv1d(skinny): %s", hipGetErrorString(e));
        *rc = ALIVE_ERR_LAUNCH;
    } else {
        *rc = ALIVE_OK;
    }
    return true;
}
