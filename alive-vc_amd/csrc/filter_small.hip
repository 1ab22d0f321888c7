// FilterBlock.forward for the 16- and 8-channel scales of the decoder's U-Net
// (/root/reference/module/decoder.py:105-150): input_conv 1x1 + 3 x FilterResBlock, i.e. six GELU -> FiLM ->
// reflect-left causal k5 convs with dilations 1,1,2,2,4,4 and three residual adds, fused into ONE kernel.
//
// At 72 000 / 144 000 samples x 16 / 8 channels these layers are HBM-bound when run conv by conv (each conv
// reads and writes 1-2 full tensors: ~14 tensor passes per scale).  Here a block owns a time tile, keeps the
// three live tensors (residual stream h, modulated conv input z, intermediate y) of that tile in LDS, and
// recomputes the 56-sample causal halo (4 x (1+1+2+2+4+4)) instead of exchanging it: HBM traffic drops to the
// input, the U-Net skip and the output.  Arithmetic is plain fp32 FMA on the VALU (C <= 16 makes an MFMA
// tile at least half empty); every thread owns fixed time columns through all seven convs, weights and the
// FiLM rows of the tile are broadcast from LDS.
#include "conv_epilogue.h"

namespace {

constexpr int HALO = 56;
constexpr int NCONV = 6;
constexpr int NFP = 8;          // FiLM frames staged per tile (tile span / 160 or / 320 + 2 taps)

template <int C>
struct SmallCfg {
    static constexpr int SPT = C == 8 ? 4 : 2;          // columns per thread
    static constexpr int BL = 256 * SPT;                // columns per tile incl. halo
    static constexpr int TT = BL - HALO;                // output columns per tile
    static constexpr int WFLOATS = C * C + C + NCONV * (5 * C * C + C);
};

template <int C>
__global__ __launch_bounds__(256, 1) void filter_block_small_kernel(const float* __restrict__ U, int L,
                                                                    const float* __restrict__ wpack,
                                                                    const float* __restrict__ film, int film_rows, int Lf,
                                                                    int film_off, float ratio, const float* __restrict__ skip,
                                                                    float* __restrict__ out) {
    using Cfg = SmallCfg<C>;
    constexpr int SPT = Cfg::SPT, BL = Cfg::BL, TT = Cfg::TT;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* bufH = sm;                       // [C][BL]
    float* bufZ = bufH + C * BL;            // [C][BL]
    float* bufY = bufZ + C * BL;            // [C][BL]
    float* W = bufY + C * BL;               // packed weights
    float* Fs = W + Cfg::WFLOATS;           // [NCONV][2][C][NFP]

    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * TT;
    const int tbase = t0 - HALO;
    const float* Un = U + (size_t)n * C * L;

    for (int e = tid; e < Cfg::WFLOATS; e += 256) W[e] = wpack[e];
    // FiLM rows of the tile: frames covering columns [max(tbase,0), t0+TT)
    int f_lo;
    {
        int ta = tbase < 0 ? 0 : tbase;
        f_lo = lerp_coord(ta, ratio, Lf).i0;
    }
    for (int e = tid; e < NCONV * 2 * C * NFP; e += 256) {
        int f = e % NFP, c = (e / NFP) % C, sel = (e / (NFP * C)) & 1, q = e / (NFP * C * 2);
        int fr = f_lo + f;
        fr = fr < Lf ? fr : Lf - 1;
        Fs[e] = film[((size_t)n * film_rows + film_off + q * 2 * C + sel * C + c) * Lf + fr];
    }
    // per-column interpolation taps (same for every conv): column i = tid + 256*s, t = tbase + i
    int li0[SPT], li1[SPT];
    float lw0[SPT], lw1[SPT];
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        int t = tbase + tid + 256 * s;
        t = t < 0 ? 0 : (t < L ? t : L - 1);
        Lerp lp = lerp_coord(t, ratio, Lf);
        li0[s] = lp.i0 - f_lo;
        li1[s] = lp.i1 - f_lo;
        if (li0[s] > NFP - 1) li0[s] = NFP - 1;
        if (li1[s] > NFP - 1) li1[s] = NFP - 1;
        lw0[s] = lp.w0;
        lw1[s] = lp.w1;
    }
    // stage the input tile
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        const int i = tid + 256 * s, t = tbase + i;
        const bool ok = t >= 0 && t < L;
#pragma unroll
        for (int c = 0; c < C; ++c) bufZ[c * BL + i] = ok ? Un[(size_t)c * L + t] : 0.0f;
    }
    __syncthreads();

    auto modulate = [&](int q, int c, int s, float v) {       // gelu -> FiLM of conv q's input
        const float* f = Fs + ((q * 2) * C + c) * NFP;
        float sc = fmaf(lw0[s], f[li0[s]], lw1[s] * f[li1[s]]);
        float sh = fmaf(lw0[s], f[C * NFP + li0[s]], lw1[s] * f[C * NFP + li1[s]]);
        return gelu_fast(v) * sc + sh;
    };

    // ---- input_conv (1x1): h = Win * U + b ; z0 = mod_0(h)     (decoder.py:147) ----
    {
        const float* Win = W;               // [ci][co]
        const float* bin = W + C * C;
        float acc[C][SPT];
#pragma unroll
        for (int co = 0; co < C; ++co)
#pragma unroll
            for (int s = 0; s < SPT; ++s) acc[co][s] = bin[co];
#pragma unroll
        for (int ci = 0; ci < C; ++ci) {
            float x[SPT];
#pragma unroll
            for (int s = 0; s < SPT; ++s) x[s] = bufZ[ci * BL + tid + 256 * s];
#pragma unroll
            for (int co = 0; co < C; ++co) {
                float w = Win[ci * C + co];
#pragma unroll
                for (int s = 0; s < SPT; ++s) acc[co][s] = fmaf(w, x[s], acc[co][s]);
            }
        }
#pragma unroll
        for (int co = 0; co < C; ++co)
#pragma unroll
            for (int s = 0; s < SPT; ++s) {
                const int i = tid + 256 * s;
                bufH[co * BL + i] = acc[co][s];
                bufZ[co * BL + i] = modulate(0, co, s, acc[co][s]);      // same column, all ci already consumed
            }
    }
    __syncthreads();

    // ---- three FilterResBlocks: q = 2j (c1), 2j+1 (c2), dilation 2^j    (decoder.py:128-134) ----
    const float* Wq = W + C * C + C;
#pragma unroll 1
    for (int q = 0; q < NCONV; ++q) {
        const int d = 1 << (q >> 1);
        const bool second = q & 1;
        const float* in = second ? bufY : bufZ;
        const float* wq = Wq + q * (5 * C * C + C);       // [ci][j][co]
        const float* bq = wq + 5 * C * C;
        float acc[C][SPT];
#pragma unroll
        for (int co = 0; co < C; ++co)
#pragma unroll
            for (int s = 0; s < SPT; ++s) acc[co][s] = bq[co];
#pragma unroll 1
        for (int ci = 0; ci < C; ++ci) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                float x[SPT];
#pragma unroll
                for (int s = 0; s < SPT; ++s) {
                    int t = tbase + tid + 256 * s + (j - 4) * d;
                    t = t < 0 ? -t : t;                                  // ReflectionPad1d on the left (common.py:88)
                    int i = t - tbase;
                    i = i < BL ? i : BL - 1;                             // only garbage columns can get here
                    x[s] = in[ci * BL + i];
                }
                const float* w = wq + (ci * 5 + j) * C;
#pragma unroll
                for (int co = 0; co < C; ++co) {
                    float wv = w[co];
#pragma unroll
                    for (int s = 0; s < SPT; ++s) acc[co][s] = fmaf(wv, x[s], acc[co][s]);
                }
            }
        }
        float* dst = second ? bufZ : bufY;
#pragma unroll
        for (int co = 0; co < C; ++co)
#pragma unroll
            for (int s = 0; s < SPT; ++s) {
                const int i = tid + 256 * s;
                float v = acc[co][s];
                if (second) {
                    v = v + bufH[co * BL + i];
                    bufH[co * BL + i] = v;
                }
                if (q + 1 < NCONV) dst[co * BL + i] = modulate(q + 1, co, s, v);
            }
        __syncthreads();
    }

    // ---- store the tile (+ U-Net skip, decoder.py:191) ----
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        const int i = tid + 256 * s, t = tbase + i;
        if (i >= HALO && t < L) {
#pragma unroll
            for (int co = 0; co < C; ++co) {
                const size_t o = ((size_t)n * C + co) * L + t;
                float v = bufH[co * BL + i];
                if (skip != nullptr) v = v + skip[o];
                out[o] = v;
            }
        }
    }
}

template <int C>
int launch_small(const float* U, int N, int L, const float* wpack, const float* film, int film_rows, int Lf, int film_off,
                 const float* skip, float* out, hipStream_t s) {
    using Cfg = SmallCfg<C>;
    const int lds = (3 * C * Cfg::BL + Cfg::WFLOATS + NCONV * 2 * C * NFP) * (int)sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)filter_block_small_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            alive_set_error("alive_filter_block_small: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e));
            return ALIVE_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const float ratio = (float)Lf / (float)L;
    ALIVE_CHECK_ARG((double)Cfg::BL * Lf / L + 3.0 <= NFP, "alive_filter_block_small: tile spans more than %d frames (L %d, Lf %d)", NFP, L, Lf);
    dim3 g(cdiv(L, Cfg::TT), N);
    filter_block_small_kernel<C><<<g, 256, lds, s>>>(U, L, wpack, film, film_rows, Lf, film_off, ratio, skip, out);
    ALIVE_CHECK_LAUNCH("alive_filter_block_small");
    return ALIVE_OK;
}

}  // namespace

extern "C" int alive_filter_block_small_weights(int C) {
    return C == 8 ? SmallCfg<8>::WFLOATS : (C == 16 ? SmallCfg<16>::WFLOATS : -1);
}

extern "C" int alive_filter_block_small(const float* U, int N, int C, int L, const float* wpack, const float* film,
                                        int film_rows, int Lf, int film_off, const float* skip, float* out, void* stream) {
    ALIVE_CHECK_ARG(U && wpack && film && out, "alive_filter_block_small: null pointer");
    ALIVE_CHECK_ARG(N > 0 && L > 16 && Lf > 0, "alive_filter_block_small: bad sizes (L must exceed the largest reflect pad, 16)");
    ALIVE_CHECK_ARG(C == 8 || C == 16, "alive_filter_block_small: C must be 8 or 16, got %d", C);
    ALIVE_CHECK_ARG(U != out, "alive_filter_block_small: in-place not supported (tiles read a halo of their left neighbour)");
    if (C == 8) return launch_small<8>(U, N, L, wpack, film, film_rows, Lf, film_off, skip, out, (hipStream_t)stream);
    return launch_small<16>(U, N, L, wpack, film, film_rows, Lf, film_off, skip, out, (hipStream_t)stream);
}
