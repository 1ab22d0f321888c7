// FilterBlock.forward for the 16- and 8-channel scales of the decoder's U-Net
// (/root/reference/module/decoder.py:105-150): input_conv 1x1 + 3 x FilterResBlock, i.e. six GELU -> FiLM ->
// reflect-left causal k5 convs with dilations 1,1,2,2,4,4 and three residual adds, fused into ONE kernel.
//
// At 72 000 / 144 000 samples x 16 / 8 channels these layers are HBM-bound when run conv by conv (each conv
// reads and writes 1-2 full tensors: ~14 tensor passes per scale).  Here a block owns a time tile and keeps it on
// chip through all seven convs: the modulated conv input z and the intermediate y live in two LDS buffers, the
// residual stream h in registers; the 56-sample causal halo (4 x (1+1+2+2+4+4)) is recomputed instead of
// exchanged, so HBM traffic drops to the input, the U-Net skip and the output.
//
// Arithmetic: exact fp32 on v_mfma_f32_16x16x4_f32 with the WEIGHTS stationary in registers: a conv has
// K = 5*C <= 80, i.e. at most 20 MFMA k-steps, so the A fragments of all seven convs are 124 VGPRs per lane for
// the whole kernel, and the only per-MFMA operand traffic is one ds_read_b32 of the B fragment (16 time columns x
// 4 k).  (A first VALU version streamed the weights through SGPRs / LDS and was bound by those fetches: 4.6 ms vs
// the ~0.6 ms of MFMA work.)  A wave owns fixed 16-column groups through all convs, so the epilogue of a group
// (residual add, GELU, FiLM) is register-local and only the modulated result goes back to LDS.
#include "conv_epilogue.h"

namespace {

#ifdef ALIVE_STAMPS
__device__ long long* g_small_stamps = nullptr;
#endif

constexpr int HALO = 56;
constexpr int NCONV = 6;
constexpr int NFP = 10;         // FiLM frames staged per tile (tile span / 160 or / 320 + 2 taps + slack)
constexpr int NT = 512;         // 8 waves: two per SIMD

template <int C>
struct SmallCfg {
    static constexpr int BL = C == 8 ? 768 : 1024;      // columns per tile incl. halo.  At 8 channels the tile buffers are small
                                                        // enough for the blocks per CU to matter more than the halo (64 windows
                                                        // x 144 000 samples: 256 cols 1.83 ms, 512 1.67, 768 1.62, 1024 1.83; at 16
                                                        // channels 1024 stays best: 1.32 against 1.42 / 1.52 ms)
    static constexpr int TT = BL - HALO;                // output columns per tile
    static constexpr int P = BL + 16;                   // LDS row pitch: the four k-rows of a B fragment land on disjoint banks
    static constexpr int NG = BL / 16;                  // 16-column groups per tile
    static constexpr int G = NG / 8;                    // groups per wave
    static constexpr int KS = 5 * C / 4;                // MFMA k-steps of a k5 conv (20 / 10)
    static constexpr int KS_IN = C / 4;                 // k-steps of the 1x1 input conv
    static constexpr int WFLOATS = (C + NCONV * 5 * C) * 16 + (1 + NCONV) * 16;   // [k][16 co] per conv, then 7 x bias[16]
};

template <int C>
__global__ __launch_bounds__(NT, 2) void filter_block_small_kernel(const float* __restrict__ U, int L,
                                                                   const float* __restrict__ wpack,
                                                                   const float* __restrict__ film, int film_rows, int Lf,
                                                                   int film_off, float ratio, int t_off, int f_off, int film_ld, const float* __restrict__ skip,
                                                                   float* __restrict__ out) {
    using Cfg = SmallCfg<C>;
    constexpr int BL = Cfg::BL, TT = Cfg::TT, P = Cfg::P, G = Cfg::G, KS = Cfg::KS, KS_IN = Cfg::KS_IN;
#ifdef ALIVE_STAMPS                 // diagnostic build only (make EXTRA=-DALIVE_STAMPS; tools/bench_filter_small.py)
#define FS_STAMP(i) ts[i] = wall_clock64()
    long long ts[5];
#else
#define FS_STAMP(i)
#endif
    FS_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* bufZ = sm;                       // [C][P]
    float* bufY = bufZ + C * P;             // [C][P]
    float* Fs = bufY + C * P;               // [NCONV][2][C][NFP]
    uint2* Xc = (uint2*)(Fs + NCONV * 2 * C * NFP);   // [BL] interpolation coordinates of a column: (i0 | i1 << 16, w1)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ln = lane & 15, lq = lane >> 4;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * TT;
    const int tbase = t0 - HALO;
    const float* Un = U + (size_t)n * C * L;

    // ---- stationary A fragments: lane (co = ln, k-slot lq) of k-step s holds W[k = 4s + lq][co] ----
    // (the input conv's here; each k5 conv reloads its 20 / 10 values from L2 at the top of its pass, which keeps the
    //  register footprint at two waves per SIMD and the conv loop rolled)
    float a_in[KS_IN], bias_in[4];
#pragma unroll
    for (int s = 0; s < KS_IN; ++s) a_in[s] = wpack[(4 * s + lq) * 16 + ln];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_in[r] = wpack[(C + NCONV * 5 * C) * 16 + lq * 4 + r];

    // ---- FiLM rows of the tile ----
    const int f_lo = lerp_coord((tbase < 0 ? 0 : tbase) + t_off, ratio, Lf).i0;     // frames of the WINDOW (t_off: range mode)
    for (int e = tid; e < NCONV * 2 * C * NFP; e += NT) {
        int f = e % NFP, c = (e / NFP) % C, sel = (e / (NFP * C)) & 1, q = e / (NFP * C * 2);
        int fr = f_lo + f;
        fr = fr < Lf ? fr : Lf - 1;
        int fc = fr - f_off;                             // frame of the window -> column of the film tensor
        fc = fc < 0 ? 0 : (fc < film_ld ? fc : film_ld - 1);
        Fs[e] = film[((size_t)n * film_rows + film_off + q * 2 * C + sel * C + c) * film_ld + fc];
    }
    for (int i = tid; i < BL; i += NT) {
        int t = tbase + i;
        t = t < 0 ? 0 : (t < L ? t : L - 1);
        const Lerp lp = lerp_coord(t + t_off, ratio, Lf);
        int i0 = lp.i0 - f_lo, i1 = lp.i1 - f_lo;
        i0 = i0 < NFP - 1 ? i0 : NFP - 1;
        i1 = i1 < NFP - 1 ? i1 : NFP - 1;
        Xc[i] = make_uint2((unsigned)i0 | ((unsigned)i1 << 16), __float_as_uint(lp.w1));
    }
    // ---- stage the input tile (raw U) into bufZ: 16-B vectors, all of a thread's loads in flight together ----
    // (tbase, L and the row pitch are multiples of 4, so a vector lies entirely inside or outside [0, L))
    {
        constexpr int NVEC = C * (BL / 4) / NT;          // 8 (C = 16) or 4 (C = 8) vectors per thread
        f32x4 v[NVEC];
#pragma unroll
        for (int k = 0; k < NVEC; ++k) {
            const int e = tid + NT * k;
            const int c = e / (BL / 4), i4 = (e - c * (BL / 4)) * 4;
            const int t = tbase + i4;
            v[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (t >= 0 && t + 3 < L) v[k] = *(const f32x4*)(Un + (size_t)c * L + t);
            else if (t >= 0 && t < L) { for (int q = 0; q < 4 && t + q < L; ++q) v[k][q] = Un[(size_t)c * L + t + q]; }
        }
#pragma unroll
        for (int k = 0; k < NVEC; ++k) {
            const int e = tid + NT * k;
            const int c = e / (BL / 4), i4 = (e - c * (BL / 4)) * 4;
            *(f32x4*)&bufZ[c * P + i4] = v[k];
        }
    }
    __syncthreads();

    // gelu -> FiLM of conv q's input, for channel rows lq*4 + r of column `col`; two channels per instruction on the
    // packed fp32 pipe, interpolation coordinates from the per-column table
    auto modulate_store = [&](int q, float* dst, int col, const f32x4& v) {
        const uint2 xc = Xc[col];
        const int i0 = xc.x & 0xffff, i1 = xc.x >> 16;
        const float w1 = __uint_as_float(xc.y), w0 = 1.0f - w1;
        if (lq * 4 >= C) return;                         // C = 8: the upper half of the MFMA tile is padding
        const float* f = Fs + ((q * 2) * C + lq * 4) * NFP;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float* fa = f + (2 * e) * NFP;
            const float* fb = fa + NFP;
            const f32x2 s0 = {fa[i0], fb[i0]}, s1 = {fa[i1], fb[i1]};
            const f32x2 h0 = {fa[C * NFP + i0], fb[C * NFP + i0]}, h1 = {fa[C * NFP + i1], fb[C * NFP + i1]};
            const f32x2 sc = pk_fma(pk_splat(w0), s0, pk_splat(w1) * s1);          // fma(w0, a, round(w1 * b))
            const f32x2 sh = pk_fma(pk_splat(w0), h0, pk_splat(w1) * h1);
            const f32x2 z = gelu_fast2(f32x2{v[2 * e], v[2 * e + 1]}) * sc + sh;
            dst[(lq * 4 + 2 * e) * P + col] = z[0];
            dst[(lq * 4 + 2 * e + 1) * P + col] = z[1];
        }
    };

    // B-fragment row offsets: k = 4s + lq -> ci = (4s + lq) % C = 4*(s % (C/4)) + lq
    int rowoff[C / 4];
#pragma unroll
    for (int u = 0; u < C / 4; ++u) rowoff[u] = (4 * u + lq) * P;

    FS_STAMP(1);
    // residual stream of this wave's column groups (group g = wv + 8*i, column = 16 g + ln), MFMA C layout
    f32x4 h[G];

    // ---- input_conv (1x1): h = Win * U + b ; z0 = mod_0(h), written back over the same columns ----
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int col = (wv + 8 * i) * 16 + ln;
        f32x4 acc = {bias_in[0], bias_in[1], bias_in[2], bias_in[3]};
#pragma unroll
        for (int s = 0; s < KS_IN; ++s)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_in[s], bufZ[rowoff[s] + col], acc, 0, 0, 0);
        h[i] = acc;
    }
    __syncthreads();            // every wave has read its raw columns (a group's rows are read by that group only, but
                                // the A x B product of one MFMA gathers all 16 lanes' columns: keep it simple and safe)
#pragma unroll
    for (int i = 0; i < G; ++i) modulate_store(0, bufZ, (wv + 8 * i) * 16 + ln, h[i]);
    __syncthreads();

    FS_STAMP(2);
    // ---- three FilterResBlocks: q = 2j (c1), 2j+1 (c2), dilation 2^j    (decoder.py:128-134) ----
#pragma unroll 1
    for (int q = 0; q < NCONV; ++q) {
        float a_q[KS], bias_q[4];
#pragma unroll
        for (int s = 0; s < KS; ++s) a_q[s] = wpack[(C + q * 5 * C + 4 * s + lq) * 16 + ln];
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_q[r] = wpack[(C + NCONV * 5 * C) * 16 + (1 + q) * 16 + lq * 4 + r];
        const int d = 1 << (q >> 1);
        const bool second = q & 1;
        const float* in = second ? bufY : bufZ;
        float* dst = second ? bufZ : bufY;
        // B fragments of a group (KS values per lane) are read one group AHEAD of their MFMAs: with the weights in
        // registers the ds_read latency is the only thing an MFMA could wait for.
        auto load_group = [&](int i, float (&bv)[KS]) {
            const int col = (wv + 8 * i) * 16 + ln;
            int idx[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                int ta = tbase + col + (j - 4) * d;
                ta = ta < 0 ? -ta : ta;                      // ReflectionPad1d on the left (common.py:88)
                int ia = ta - tbase;
                ia = ia < 0 ? 0 : ia;                        // only never-stored halo columns can be clamped
                idx[j] = ia < BL ? ia : BL - 1;
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) bv[s] = in[rowoff[s % (C / 4)] + idx[(4 * s) / C]];
        };
        float bcur[KS], bnxt[KS];
        load_group(0, bcur);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (i + 1 < G) load_group(i + 1, bnxt);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc0 = {bias_q[0], bias_q[1], bias_q[2], bias_q[3]};
            f32x4 acc1 = {0.0f, 0.0f, 0.0f, 0.0f};           // two chains: the 16x16x4 MFMA has a 40-cycle dependent latency
#pragma unroll
            for (int s = 0; s < KS; s += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_q[s], bcur[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_q[s + 1], bcur[s + 1], acc1, 0, 0, 0);
            }
            acc0 = acc0 + acc1;
            if (second) {
                acc0 = acc0 + h[i];
                h[i] = acc0;
            }
            if (q + 1 < NCONV) modulate_store(q + 1, dst, (wv + 8 * i) * 16 + ln, acc0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS; ++s) bcur[s] = bnxt[s];
        }
        __syncthreads();
    }

    FS_STAMP(3);
    // ---- store the tile (+ U-Net skip, decoder.py:191): through LDS, so that global accesses are 16-B vectors along t and
    // all skip vectors of a thread are in flight together ----
    float* Ht = bufZ;                                    // [C][P] fp32: both conv buffers are free now
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int col = (wv + 8 * i) * 16 + ln;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (lq * 4 + r < C) Ht[(lq * 4 + r) * P + col] = h[i][r];
    }
    __syncthreads();
    {
        constexpr int NV = (C * (TT / 4) + NT - 1) / NT;     // 242 vectors per channel row
        f32x4 sk[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e = tid + NT * k;
            const int c = e / (TT / 4), t = t0 + (e - c * (TT / 4)) * 4;
            sk[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (skip != nullptr && e < C * (TT / 4) && t + 3 < L) sk[k] = *(const f32x4*)(skip + ((size_t)n * C + c) * L + t);
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e = tid + NT * k;
            const int c = e / (TT / 4), c4 = (e - c * (TT / 4)) * 4, t = t0 + c4;
            if (e >= C * (TT / 4) || t >= L) continue;
            const size_t o = ((size_t)n * C + c) * L + t;
            const f32x4 v = *(const f32x4*)&Ht[c * P + HALO + c4];
            if (t + 3 < L) {
                *(f32x4*)(out + o) = v + sk[k];
            } else {
                for (int q = 0; q < 4 && t + q < L; ++q) out[o + q] = v[q] + (skip != nullptr ? skip[o + q] : 0.0f);
            }
        }
    }
#ifdef ALIVE_STAMPS
    if (g_small_stamps != nullptr && tid == 0) {
        long long* o = g_small_stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
        o[0] = ts[1] - ts[0]; o[1] = ts[2] - ts[1]; o[2] = ts[3] - ts[2]; o[3] = wall_clock64() - ts[3];
    }
#endif
}

template <int C>
int launch_small(const float* U, int N, int L, const float* wpack, const float* film, int film_rows, int Lf, int film_off,
                 int t0, int f0, int film_ld, const float* skip, float* out, hipStream_t s) {
    using Cfg = SmallCfg<C>;
    const int lds = (2 * C * Cfg::P + NCONV * 2 * C * NFP) * (int)sizeof(float) + Cfg::BL * 8;
    {
        static LdsOptIn optin;                               // one per instantiation <C>
        hipError_t e = optin.ensure({(const void*)filter_block_small_kernel<C>}, lds);
        if (e != hipSuccess) {
            alive_set_error("alive_filter_block_small: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e));
            return ALIVE_ERR_LAUNCH;
        }
    }
    const float ratio = (float)film_ld / (float)L;       // == window frames / window samples at this rate
    ALIVE_CHECK_ARG((double)Cfg::BL * film_ld / L + 3.0 <= NFP, "alive_filter_block_small: tile spans more than %d frames (L %d, frames %d)", NFP, L, film_ld);
    dim3 g(cdiv(L, Cfg::TT), N);
    filter_block_small_kernel<C><<<g, NT, lds, s>>>(U, L, wpack, film, film_rows, Lf, film_off, ratio, t0, f0, film_ld, skip, out);
    ALIVE_CHECK_LAUNCH("alive_filter_block_small");
    return ALIVE_OK;
}

}  // namespace

extern "C" int alive_filter_block_small_weights(int C) {
    return C == 8 ? SmallCfg<8>::WFLOATS : (C == 16 ? SmallCfg<16>::WFLOATS : -1);
}

extern "C" int alive_filter_block_small(const float* U, int N, int C, int L, const float* wpack, const float* film,
                                        int film_rows, int Lf, int film_off, const float* skip, float* out, void* stream) {
    return alive_filter_block_small_range(U, N, C, L, wpack, film, film_rows, Lf, film_off, 0, 0, Lf, skip, out, stream);
}

extern "C" int alive_filter_block_small_range(const float* U, int N, int C, int L, const float* wpack, const float* film,
                                              int film_rows, int Lf, int film_off, int t0, int f0, int film_ld,
                                              const float* skip, float* out, void* stream) {
    ALIVE_CHECK_ARG(U && wpack && film && out, "alive_filter_block_small: null pointer");
    ALIVE_CHECK_ARG(N > 0 && L > 16 && Lf > 0, "alive_filter_block_small: bad sizes (L must exceed the largest reflect pad, 16)");
    ALIVE_CHECK_ARG(C == 8 || C == 16, "alive_filter_block_small: C must be 8 or 16, got %d", C);
    ALIVE_CHECK_ARG(U != out, "alive_filter_block_small: in-place not supported (tiles read a halo of their left neighbour)");
    ALIVE_CHECK_ARG((L & 3) == 0 && ((((uintptr_t)U) | ((uintptr_t)out) | ((uintptr_t)skip)) & 15) == 0,
                    "alive_filter_block_small: L must be a multiple of 4 and U / out / skip 16-byte aligned");
    ALIVE_CHECK_ARG(film_ld > 0 && t0 >= 0 && f0 >= 0, "alive_filter_block_small: bad frame range");
    if (C == 8) return launch_small<8>(U, N, L, wpack, film, film_rows, Lf, film_off, t0, f0, film_ld, skip, out, (hipStream_t)stream);
    return launch_small<16>(U, N, L, wpack, film, film_rows, Lf, film_off, t0, f0, film_ld, skip, out, (hipStream_t)stream);
}

#ifdef ALIVE_STAMPS
extern "C" void alive_debug_set_stamps_small(long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_small_stamps), &p, sizeof(p)); }
#endif
