// FilterBlock.forward for the 256- and the 64-channel scale of the decoder's U-Net (/root/reference/module/decoder.py:105-150; the 1x1
// input conv is composed into the transposed conv in front of the block, module/_pack.py), fused into ONE kernel on plain fp16 operands
// (decoder precision mode 1, batch path): round 6.  DESIGN.md 3.2g has the measurements behind every choice below.
//
// Conv by conv (conv_split_kernel<128, 1, true, true>, six launches + alive_gelu_film per window batch) every k5 conv of the 256-channel
// block read one fp16 plane and wrote one, every second one also streamed the fp32 residual in and out, and each launch ran at ~0.2 of
// the matrix pipe.  Here a block keeps a tile -- 128 columns x 256 channels or 512 x 64: 64 KB -- on chip through all six
// GELU -> FiLM -> reflect-left causal k5 convs (dilations 1, 1, 2, 2, 4, 4):
//   * the modulated conv input lives in two LDS buffers as ONE fp16 plane, [column][C channels], 16-byte chunks XOR-swizzled per row
//     (Geo::swz): a 32x32x16 B fragment is one conflict-free ds_read_b128 at any tap shift;
//   * a wave owns 32 RG output channels x 128 columns: a k-step (one tap x 16 channels) is RG A fragments of the weights, streamed from
//     L2 in the k-blocked fp16 slab of module/_pack.py::pack_conv_split_h -- each fragment once per block and conv at C = 256 --,
//     four B fragments from LDS and 4 RG MFMAs, ONE memory instruction behind each MFMA;
//   * the fp32 residual stream of the wave's outputs stays in registers from the tile's load to its store (+ U-Net skip);
//   * the batch is ONE sequence of tiles, window after window; a block takes a run of them, every tile all-new columns, and each conv's
//     causal context -- the last 16 columns of its input -- waits for the next tile in the block's slice of a workspace (L2); a run that
//     starts inside a window begins with a warm-up tile that is not stored; at a window's first tile the context is the reflection of
//     the tile's own columns 1 .. 16 (ReflectionPad1d, common.py:88);
//   * bufY has no context rows of its own: its rows -16 .. -1 ARE bufZ's last 16 rows (the buffers are adjacent), which are dead
//     whenever bufY is read -- their content (the tail of the previous conv's input) has gone to the workspace by then;
//   * the FiLM rows a wave applies are those of its own channels: each wave stages them in a private LDS table (loads: a lane per
//     FRAME, several rows per instruction) -- no block barrier for them;
//   * every LDS access goes through address-space-3 pointers (lds_get / lds_put): see the note there.
// HBM traffic: the residual stream in, the skip in, the output out, the FiLM rows (12 B per value instead of ~48).
#include "conv_epilogue.h"
#include <stdlib.h>

namespace {

constexpr int NCONV = 6;
constexpr int CTX = 16;                  // context rows in front of a buffer (4 taps back x dilation 4)
constexpr int KW = 5;
constexpr int PF1 = 4, PF2 = 4;          // k-steps of weights in flight per wave (one / two waves per SIMD)

// The geometry of a block for C channels (256: decoder scale 0; 64: scale 1) and RG 32-channel row groups per wave.  A wave owns
// 32 RG channels x 128 columns; at C = 256 the waves tile the channels and a tile is 128 columns, at C = 64 they tile the channels (RG = 1:
// two waves) and four column groups: a tile is 512 columns.  Either way a buffer is 64 KB.
template <int C, int RG>
struct Geo {
    static constexpr int CW = 32 * RG;                       // channels per wave
    static constexpr int NWR = C / CW;                       // waves along the channels
    static constexpr int NWC = C == 256 ? 1 : 4;             // column groups of waves
    static constexpr int NW = NWR * NWC, NT = 64 * NW;
    static constexpr int BL = 128 * NWC;                     // columns per tile
    static constexpr int ROWB = 2 * C;                       // bytes per LDS row: one column, all channels, fp16
    static constexpr int CPR = ROWB / 16;                    // 16-byte chunks (8 channels) per row
    static constexpr int GUARD = CTX * ROWB;                 // a conv's context: 8 KB / 2 KB
    static constexpr int BUFB = BL * ROWB;                   // 64 KB
    static constexpr int LG = C == 256 ? 16 : 8;             // lanes (= frames) per FiLM row in a staging load; the frames a wave's table holds
    static constexpr int NFS = LG;                           //   (128 columns span <= 12.8 frames at 10 samples per frame, 1.6 at 80; + 1, + slack)
    static constexpr int FS_WAVE = CW / 2 * NFS * 16;        // bytes of a wave's table: [CW / 2 channel pairs][NFS] x (scale / 2 of both, shift of both)
    // One region in front of bufZ serves three tenants in turn, conv by conv: the waves' bias lines (its start: from a conv's start to the
    // accumulators' initialisation), bufZ's 16 context rows (its end: from there to the end of the k-loop, even convs), the waves' FiLM
    // tables (all of it: from the barrier behind the k-loop to the end of the epilogue).  At C = 256 with bufZ and bufY: all 160 KB.
    static constexpr int FS_BYTES = NW * FS_WAVE;            // 32 KB / 16 KB
    static constexpr int SLOTS = C == 64 ? NCONV * GUARD : 0;   // C = 64: the convs' contexts wait for the next tile in LDS (12 KB); C = 256: workspace
    static constexpr int LDS_BYTES = FS_BYTES + 2 * BUFB + SLOTS;
    static constexpr int NKC = C / 16;                       // k-steps per tap
    static constexpr int NKS = KW * NKC;                     // 80 / 20 k-steps per conv
    static_assert(FS_BYTES >= GUARD && FS_BYTES >= NW * CW * 4 && NKC % 2 == 0, "region in front of bufZ");
    // the 16-byte chunk c of row r sits at position c ^ swz(r): 16 lanes of a ds_read_b128 group (16 rows, one chunk index) then hit 16
    // different bank quads -- rows 512 B apart share their banks (swizzle on the row's low 4 bits), rows 128 B apart alternate between
    // two halves of them (swizzle on bits 1 .. 3)
    static __device__ __forceinline__ int swz(int row) { return C == 256 ? (row & 15) : ((row >> 1) & 7); }
};

// LDS by byte offset, through pointers of the LDS address space only.  (As generic pointers -- selected between the two buffers, captured
// by the lambdas -- hipcc 7.2 at times leaves one as a flat pointer with a null test it cannot select: "Illegal instruction detected:
// V_CMP_NE_U32_e32 0, $src_shared_base", on and off with unrelated edits.)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"      // (the host pass sees 64-bit pointers; LDS pointers are 32 bits)
// (`off`: a compile-time constant where there is one -- added in the pointer domain it folds into the instruction's offset field)
template <class T>
__device__ __forceinline__ T lds_get(int base, int off = 0) {
    return *(const __attribute__((address_space(3))) T*)((const __attribute__((address_space(3))) unsigned char*)(unsigned)base + off);
}
template <class T>
__device__ __forceinline__ void lds_put(int base, int off, T v) {
    *(__attribute__((address_space(3))) T*)((__attribute__((address_space(3))) unsigned char*)(unsigned)base + off) = v;
}
#pragma clang diagnostic pop
#ifdef ALIVE_FB256_PROF        // (phase clocks of block 7, printed per call: make EXTRA=-DALIVE_FB256_PROF; tools/bench_fb256.py)
__device__ unsigned long long fb256_prof[16];
#define PROF(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define PROF(i) do {} while (0)
#endif


struct Fb256Weights {
    const unsigned short* w[NCONV];      // fp16 slab of each conv, k-blocked [K / 32][C][32], K = 5 x C tap-major
    const float* b[NCONV];
};

// RG: 32-channel row groups per wave.  2: four waves of 64 channels x 128 columns (one per SIMD, 512 registers).  1: eight waves of 32
// channels x 128 columns (two per SIMD, 256 registers): every weight fragment is still loaded once per block (eight waves of 64 channels
// x 64 columns load each twice -- the second read hits the L1, whose 64 B per clock then bound the k-loop: 505 cycles per k-step
// against 318, measured), the B fragments are read from LDS by twice as many waves (half of what it delivers), the epilogue's vector
// work issues from two waves per SIMD.
template <int C, int RG>
__global__ __launch_bounds__((Geo<C, RG>::NT), 1) void filter_block256_kernel(const float* __restrict__ U, int L, Fb256Weights wts, const float* __restrict__ film,
                                                                 int film_rows, int Lf, int film_off, float ratio, int t_off, int f_off,
                                                                 int film_ld, const float* __restrict__ skip, float* __restrict__ out,
                                                                 int tiles, int per_block, int total, unsigned char* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int sm0 = (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sm;      // (0: the kernel has no static LDS)
    typedef Geo<C, RG> G;
    constexpr int NT = G::NT, CTW = 4, CW = G::CW, PF = RG == 1 ? PF2 : PF1, BL = G::BL, ROWB = G::ROWB, GUARD = G::GUARD, BUFB = G::BUFB,
                  NFS = G::NFS, LG = G::LG, NKS = G::NKS, NKC = G::NKC, CPR = G::CPR;
    auto swz = [](int row) { return G::swz(row); };
    const int bufZ = sm0 + G::FS_BYTES;
    const int bufY = bufZ + BUFB;
    const int slots = bufY + BUFB;                              // (C = 64 only)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv % G::NWR;                                  // channels CW w .. CW w + CW - 1
    const int col0 = 128 * (wv / G::NWR);                       // columns col0 .. col0 + 127 of the tile
    const int Fs = sm0 + wv * G::FS_WAVE;                       // this wave's FiLM table
    const int Bs = sm0 + 4 * CW * wv;                           // its biases of the current conv
    // The batch is one sequence of tiles, window after window; a block takes per_block consecutive ones (any number of windows, any
    // place inside one) -- every CU gets the same count whatever the batch size, and one warm-up tile per block is all the redundancy.
    const int g0 = (int)blockIdx.x * per_block;
    const int g1 = g0 + per_block < total ? g0 + per_block : total;
#ifdef ALIVE_FB256_PROF
    long long pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
    // each conv's causal context (its input's last 16 columns, 8 KB) waits for the next tile in the block's slice of the workspace: 32 bytes
    // per thread and conv, written and read back by the same thread (L2-resident; in registers they cost the k-loop its schedule)
    unsigned char* const ctx_ws = ws + (size_t)blockIdx.x * (NCONV * GUARD) + tid * 16;      // (+ NT * 16 for a thread's second piece)

#pragma unroll 1
    for (int g = g0 % tiles ? g0 - 1 : g0; g < g1; ++g) {
        const bool warm = g < g0;                       // a block that starts inside a window computes the tile before for its contexts only
        const int n = g / tiles;
        const int tbase = (g - n * tiles) * BL;
        const bool first = tbase == 0;                  // the window's first tile: the context is the reflection of its own columns
        const float* Un = U + (size_t)n * C * L;
        // the FiLM rows of this wave's channels 64 w .., conv 0: scale; + 256 rows: shift; + 512 rows: the next conv
        const float* film_w = film + ((size_t)n * film_rows + film_off + CW * w) * film_ld;
        int n32 = lane & 31, lh = lane >> 5;
        asm volatile("" : "+v"(n32), "+v"(lh));         // (opaque per tile: keeps hipcc from hoisting every lane-constant address out of the tile loop)
        const unsigned short* wrow = wts.w[0] + (size_t)(CW * w + n32) * 32 + 8 * lh;

        // weights: A fragment of k-step ks = 16 j + cb for row group rg: rows 64 w + 32 rg + n32, k = 256 j + 16 cb + 8 lh .. + 7
        auto a_ptr = [&](const unsigned short* Wc, int ks, int rg) {
            const int kb = C / 32 * (ks / NKC) + ((ks % NKC) >> 1);
            return (const bf16x8*)(Wc + (wrow - wts.w[0]) + ((size_t)kb * C + 32 * rg) * 32 + (ks & 1) * 16);
        };
        bf16x8 a[PF][RG];
        auto prime = [&](int q) {                       // the first PF k-steps of conv q
#pragma unroll
            for (int s = 0; s < PF; ++s)
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) a[s][rg] = *a_ptr(wts.w[q], s, rg);
        };
        prime(0);

        // F.interpolate coordinates of this lane's four columns (window frames: t_off in range mode), and the first frame of each half's table
        int ci0[CTW]; float cw1[CTW];
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const int ct = c;
            const int t = tbase + col0 + 32 * ct + n32;
            const Lerp lp = lerp_coord((t < L ? t : L - 1) + t_off, ratio, Lf);
            ci0[c] = lp.i0;
            cw1[c] = lp.w1;
        }
        const int f_lo = lerp_coord((tbase + col0 < L ? tbase + col0 : L - 1) + t_off, ratio, Lf).i0;      // of the wave's columns
        // FiLM rows qf (the modulation in front of conv qf) of the wave's 64 channels, frames f_lo .. f_lo + 15: loads, then the table.
        // A load instruction takes four rows x 16 consecutive frames (a lane per frame: one or two cache lines per row -- a lane per
        // ROW touches 64 lines per instruction, and the CU's address unit takes them one per clock: 15 k cycles per conv, measured).
        constexpr int RPI = 64 / LG, NI = 2 * CW / RPI;           // rows per staging load, loads per table
        float fr[NI];
        const int f_lane = lane % LG, r_lane = lane / LG;
        auto film_load = [&](int qf) {
            int fa = f_lo + f_lane;
            fa = fa < Lf ? fa : Lf - 1;
            int fc = fa - f_off;                                 // frame of the window -> column of the film tensor
            fc = fc < 0 ? 0 : (fc < film_ld ? fc : film_ld - 1);
            const float* p = film_w + ((size_t)qf * 2 * C + r_lane) * film_ld + fc;
#pragma unroll
            for (int i = 0; i < NI; ++i)                          // rows RPI i + r_lane of the wave's 2 CW: its channels' scale rows, then their shift rows
                fr[i] = __builtin_nontemporal_load(p + ((size_t)(RPI * i / CW) * C + RPI * i % CW) * film_ld);
        };
        auto film_put = [&]() {
#pragma unroll
            for (int i = 0; i < NI; ++i)                          // scale rows halved (exact): see filter_mid.hip
                lds_put<float>(Fs + 4 * (((r_lane >> 1) * NFS + f_lane) * 4 + (r_lane & 1)), 4 * ((RPI * i % CW / 2) * NFS * 4 + 2 * (RPI * i / CW)),
                               RPI * i < CW ? 0.5f * fr[i] : fr[i]);
        };
        film_load(0);

        // ---- residual stream of this wave's 64 channels x 128 columns, in the MFMA C layout: h[rg][ct][4 g + e] = channel
        //      64 w + 32 rg + 8 g + 4 lh + e of column 32 ct + n32 ----
        float h[RG][CTW][16];
        auto h_get = [&](int rg, int ct, int r) { return h[rg][ct][r]; };
        auto h_set = [&](int rg, int ct, int r, float x) { h[rg][ct][r] = x; };
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                const int t = tbase + col0 + 32 * c + n32;
                const float* up = Un + (size_t)(CW * w + 32 * rg + 4 * lh) * L + (t < L ? t : L - 1);      // (columns past the end: finite, never stored)
                float x[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = __builtin_nontemporal_load(up + (size_t)(8 * (r >> 2) + (r & 3)) * L);      // (read once: not to evict the weights from L2)
#pragma unroll
                for (int r = 0; r < 16; ++r) h_set(rg, c, r, x[r]);
            }

        // GELU -> FiLM -> fp16 -> LDS of one 32 x 32 accumulator tile (channel group rg, column tile ct), FiLM rows from the wave's table
        auto emit_tile = [&](const f32x16& v, int rg, int c, int dstp) {
            int col = col0 + 32 * c + n32;
            int i0 = ci0[c] - f_lo;
            i0 = i0 < NFS - 2 ? i0 : NFS - 2;
            asm volatile("" : "+v"(col), "+v"(i0));         // (opaque per call: the 32 LDS addresses derived from them are three instructions each,
                                                            // hoisted out of the tile loop they are 100+ registers and the kernel spills)
            const float w1 = cw1[c], w0 = 1.0f - w1;
            const bool cnt = !warm && tbase + col < L;      // (saturations of stored columns only)
            float zmax = 0.0f;
            // Two adjacent channels at a time on the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two IEEE operations per issue
            // slot -- the epilogue runs with the matrix pipe idle and is bound by vector issue).  Same operations per value as the scalar
            // form of filter_mid.hip.  The table holds (scale / 2 of both channels, shift of both) per frame: one ds_read_b128 each.
            // (Four values -- two packed pairs -- per source statement: hipcc then alternates the two pairs' instructions; one pair at a
            // time it emits a dependent chain with a wait state between every two packed operations, 58 s_nop per 16 values.)
            auto fma4 = [](f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); };
            auto splat = [](float c) { return f32x4{c, c, c, c}; };
            const f32x4 W0 = splat(w0), W1 = splat(w1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int fp = Fs + 16 * (2 * lh * NFS + i0), fo = 16 * (16 * rg + 4 * g) * NFS;      // channel pair 0 of the four; pair 1: + NFS
                const f32x4 a0 = lds_get<f32x4>(fp, fo), a1 = lds_get<f32x4>(fp, fo + 16), b0 = lds_get<f32x4>(fp, fo + 16 * NFS), b1 = lds_get<f32x4>(fp, fo + 16 * NFS + 16);
                const f32x4 sc = fma4(W0, f32x4{a0[0], a0[1], b0[0], b0[1]}, W1 * f32x4{a1[0], a1[1], b1[0], b1[1]});      // ATen's linear interpolation
                const f32x4 sh = fma4(W0, f32x4{a0[2], a0[3], b0[2], b0[3]}, W1 * f32x4{a1[2], a1[3], b1[2], b1[3]});
                const f32x4 x = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                const f32x4 ax = {fabsf(x[0]), fabsf(x[1]), fabsf(x[2]), fabsf(x[3])};
                const f32x4 tq = fma4(ax, splat(0.3275911f * 0.70710678118654752440f), splat(1.0f));
                const f32x4 t = {__builtin_amdgcn_rcpf(tq[0]), __builtin_amdgcn_rcpf(tq[1]), __builtin_amdgcn_rcpf(tq[2]), __builtin_amdgcn_rcpf(tq[3])};
                const f32x4 xs = x * splat(0.84932180028801904272f);
                const f32x4 qq = xs * xs;
                const f32x4 ex = {__builtin_amdgcn_exp2f(-qq[0]), __builtin_amdgcn_exp2f(-qq[1]), __builtin_amdgcn_exp2f(-qq[2]), __builtin_amdgcn_exp2f(-qq[3])};
                f32x4 p = fma4(splat(1.061405429f), t, splat(-1.453152027f));
                p = fma4(p, t, splat(1.421413741f));
                p = fma4(p, t, splat(-0.284496736f));
                p = fma4(p, t, splat(0.254829592f));
                const f32x4 erf_abs = fma4(-(p * t), ex, splat(1.0f));
                const f32x4 z = fma4(fma4(ax, erf_abs, x), sc, sh);             // 2 gelu(x) * (scale / 2) + shift
                zmax = fmaxf(fmaxf(zmax, fmaxf(fabsf(z[0]), fabsf(z[1]))), fmaxf(fabsf(z[2]), fabsf(z[3])));
                typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                const f16x2_t h0 = {(_Float16)__builtin_amdgcn_fmed3f(z[0], -65504.0f, 65504.0f), (_Float16)__builtin_amdgcn_fmed3f(z[1], -65504.0f, 65504.0f)};
                const f16x2_t h1 = {(_Float16)__builtin_amdgcn_fmed3f(z[2], -65504.0f, 65504.0f), (_Float16)__builtin_amdgcn_fmed3f(z[3], -65504.0f, 65504.0f)};
                const int chunk = CW / 8 * w + 4 * rg + g;
                lds_put<u32x2>(dstp + col * ROWB + ((chunk ^ swz(col)) << 4) + 8 * lh, 0, u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)});
            }
            // (one count per lane and 16 values, not per converted pair: the guards read "any")
            if (__builtin_expect(cnt && zmax > 65504.0f, 0)) atomicAdd(&alive_f16_sat_count, 1u);
        };
        // the context in front of a buffer: the previous tile's (registers) or, at the window's first tile, rows 1 .. 16 reflected
        constexpr int NCH = GUARD / 16, NP = (NCH + NT - 1) / NT;      // a context's 16-byte pieces; per thread
        auto put_context = [&](int buf, int q) {
            if (first) {
#pragma unroll
                for (int u = 0; u < NP; ++u) {                  // rows 1 .. 16, CPR chunks each
                    const int e = tid + NT * u;
                    const int j = 1 + e / CPR, p = e % CPR;     // stored chunk position p in row j
                    const int c = p ^ swz(j);                   // the channel chunk it holds
                    if (NCH % NT == 0 || e < NCH) lds_put<u32x4>(buf - j * ROWB + ((c ^ swz(-j)) << 4), 0, lds_get<u32x4>(buf + j * ROWB + (p << 4)));
                }
            } else {
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    u32x4 c0 = {0u, 0u, 0u, 0u};                 // (a warm-up tile has no context: its stored columns' cone does not reach it)
                    if (NCH % NT == 0 || tid + NT * u < NCH) {
                        if (!warm) c0 = G::SLOTS ? lds_get<u32x4>(slots + q * GUARD + tid * 16, NT * 16 * u) : *(const u32x4*)(ctx_ws + q * GUARD + NT * 16 * u);
                        lds_put<u32x4>(buf - GUARD + tid * 16, NT * 16 * u, c0);
                    }
                }
            }
        };
        auto take_context = [&](int buf, int q) {          // rows BL - 16 .. BL - 1 of a conv's complete input
#pragma unroll
            for (int u = 0; u < NP; ++u)
                if (NCH % NT == 0 || tid + NT * u < NCH) {
                    const u32x4 v = lds_get<u32x4>(buf + (BL - CTX) * ROWB + tid * 16, NT * 16 * u);
                    if (G::SLOTS) lds_put<u32x4>(slots + q * GUARD + tid * 16, NT * 16 * u, v);
                    else *(u32x4*)(ctx_ws + q * GUARD + NT * 16 * u) = v;
                }
        };

        PROF(0);
        __syncthreads();                                  // the previous tile is done with the buffers
        PROF(1);
        // ---- z0 = mod_0(h) ----
        film_put();
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                f32x16 v;
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = h_get(rg, c, r);
                emit_tile(v, rg, c, bufZ);
            }
        __syncthreads();
        PROF(2);

        // ---- six convs: q = 2 j (c1), 2 j + 1 (c2), dilation 2^j   (decoder.py:128-134) ----
#pragma unroll 1
        for (int q = 0; q < NCONV; ++q) {
            const bool second = (q & 1) != 0;
            const bool last = q + 1 == NCONV;
            const int in = second ? bufY : bufZ;
            const int dst = second ? bufZ : bufY;
            const int d = 1 << (q >> 1);
            const unsigned short* Wc = wts.w[q];
            const float* bc = wts.b[q];
            // `in` is complete (barrier behind the previous stage).  Its own context goes in front of it -- the previous tile's, from the
            // registers, or the reflection -- and its tail becomes the next tile's context.  The two touch different rows: for bufY the
            // context rows are bufZ's rows 112 .. 127, which the conv before this one has finished reading (and taken).
            put_context(in, q);
            take_context(in, q);
            if (warm && last) break;                       // (a warm-up tile is computed for the contexts only: the last conv's output is nobody's)
            __syncthreads();
            PROF(3);
            f32x16 acc[RG][CTW];
            if (lane < CW) lds_put<float>(Bs + 4 * lane, 0, bc[CW * w + lane]);                  // (one coalesced load, then the C layout's 32 values per lane as 8 LDS reads)
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                f32x16 b16;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = lds_get<f32x4>(Bs + 16 * lh, 4 * (32 * rg + 8 * g));
#pragma unroll
                    for (int e = 0; e < 4; ++e) b16[4 * g + e] = v[e];
                }
#pragma unroll
                for (int c = 0; c < CTW; ++c) acc[rg][c] = b16;
            }
            // One k-step = 8 MFMAs, 4 B fragments of the NEXT k-step (LDS, into the other buffer), 2 A fragments of k-step ks + PF (L2, into
            // the ring slot this k-step has just read).  ONE memory instruction per MFMA: with a wave per SIMD a memory instruction that
            // waits for its queue (four waves in step behind the same barrier) holds up the wave's next MFMA, and four or two of them in a
            // row cost a k-step 170 of 426 cycles (measured against the loop without them); behind an MFMA the wait runs under its 32
            // cycles.  (The scheduling fences keep hipcc from regrouping them; no value outlives its register's next definition, so ring
            // and buffers stay in place.)  Each accumulator still takes one MFMA per k-step in k order: the sums are the same bits.
            auto b_frag = [&](int ks, int c) {
                const int j = ks / NKC, cb = ks % NKC;
                const int row = col0 + n32 + (j - 4) * d;                     // of column tile 0: >= -16, the context rows (32 rows on: the same swizzle)
                return lds_get<bf16x8>(in + row * ROWB + (((2 * cb + lh) ^ swz(row)) << 4), 32 * ROWB * c);
            };
            if (RG == 1 && !last) film_load(q + 1);
            bf16x8 bfr[2][CTW];
#pragma unroll
            for (int c = 0; c < CTW; ++c) bfr[0][c] = b_frag(0, c);
#pragma unroll 4
            for (int ks = 0; ks < NKS; ++ks) {
                const int k1 = ks + 1 < NKS ? ks + 1 : NKS - 1;                                // (past the end: the last k-step again, no branch)
                const int kn = ks + PF < NKS ? ks + PF : NKS - 1;
#pragma unroll
                for (int c = 0; c < CTW; ++c) {
                    acc[0][c] = mfma_f16(a[ks % PF][0], bfr[ks & 1][c], acc[0][c]);
                    __builtin_amdgcn_sched_barrier(0);
                    bfr[(ks + 1) & 1][c] = b_frag(k1, c);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (RG == 2) {
                    acc[RG - 1][0] = mfma_f16(a[ks % PF][RG - 1], bfr[ks & 1][0], acc[RG - 1][0]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                a[ks % PF][0] = *a_ptr(Wc, kn, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (RG == 2) {
#pragma unroll
                    for (int c = 1; c < CTW; ++c) acc[RG - 1][c] = mfma_f16(a[ks % PF][RG - 1], bfr[ks & 1][c], acc[RG - 1][c]);
                    __builtin_amdgcn_sched_barrier(0);
                    a[ks % PF][RG - 1] = *a_ptr(Wc, kn, RG - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            PROF(4);
            if (!last) {
                if (RG == 2) film_load(q + 1);              // (behind the k-loop: 32 registers that the loop's schedule does not have)
                prime(q + 1);                               // (in flight under the epilogue)
            }
            __syncthreads();                               // every wave is done reading `in` (and bufZ's context rows, which the tables replace)
            PROF(5);
            if (!last) {
                film_put();
            }
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int c = 0; c < CTW; ++c) {
                    f32x16 v = acc[rg][c];
                    if (second) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) { v[r] = v[r] + h_get(rg, c, r); h_set(rg, c, r, v[r]); }
                    }
                    if (!last) emit_tile(v, rg, c, dst);
                }
            __syncthreads();                               // dst complete
            PROF(6);
        }

        // ---- store (+ U-Net skip, decoder.py:191), straight from the residual registers ----
        if (!warm) {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int c = 0; c < CTW; ++c) {
                    const int t = tbase + col0 + 32 * c + n32;
                    if (t >= L) continue;
                    const size_t o = ((size_t)n * C + CW * w + 32 * rg + 4 * lh) * L + t;
                    if (skip != nullptr) {
                        float sk[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) sk[r] = __builtin_nontemporal_load(skip + o + (size_t)(8 * (r >> 2) + (r & 3)) * L);
#pragma unroll
                        for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(h_get(rg, c, r) + sk[r], out + o + (size_t)(8 * (r >> 2) + (r & 3)) * L);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(h_get(rg, c, r), out + o + (size_t)(8 * (r >> 2) + (r & 3)) * L);
                    }
                }
        }
        PROF(7);
    }
#ifdef ALIVE_FB256_PROF
    if (blockIdx.x == 7 && tid == 0)
        for (int i = 0; i < 10; ++i) atomicAdd(&fb256_prof[i], (unsigned long long)pacc[i]);
#endif
}

template <int C>
int fb_launch(const char* name, const float* U, int N, int L, const void* const* w16, const float* const* bias, const float* film, int film_rows,
              int Lf, int film_off, int t0, int f0, int film_ld, const float* skip, float* out, void* ws, int64_t ws_bytes, void* stream) {
    typedef Geo<C, 1> G;                      // (tile, context and table sizes do not depend on RG)
    ALIVE_CHECK_ARG(U && w16 && bias && film && out, "%s: null pointer", name);
    ALIVE_CHECK_ARG(N > 0 && L > 2 * CTX && Lf > 0, "%s: bad sizes (L must exceed 32)", name);
    ALIVE_CHECK_ARG(U != out, "%s: in-place not supported (a block's warm-up tile reads columns another block has stored)", name);
    ALIVE_CHECK_ARG(film_ld > 0 && t0 >= 0 && f0 >= 0, "%s: bad frame range", name);
    ALIVE_CHECK_ARG(128.0 * film_ld / L + 3.0 <= G::NFS, "%s: 128 columns span more than %d frames (L %d, frames %d)", name, G::NFS - 3, L, film_ld);
    Fb256Weights wts;
    for (int q = 0; q < NCONV; ++q) {
        ALIVE_CHECK_ARG(w16[q] && bias[q], "%s: null weights", name);
        wts.w[q] = (const unsigned short*)w16[q];
        wts.b[q] = bias[q];
    }
    {
        static LdsOptIn optin;
        hipError_t e = optin.ensure({(const void*)filter_block256_kernel<C, 2>, (const void*)filter_block256_kernel<C, 1>}, G::LDS_BYTES);
        if (e != hipSuccess) {
            alive_set_error("%s: cannot reserve %d B of LDS: %s", name, G::LDS_BYTES, hipGetErrorString(e));
            return ALIVE_ERR_LAUNCH;
        }
    }
    const float ratio = (float)film_ld / (float)L;
    const int tiles = cdiv(L, G::BL);
    ALIVE_CHECK_ARG((int64_t)N * tiles < (1ll << 31), "%s: too many tiles", name);
    static int cus = 0;
    if (cus == 0) {                              // (one process drives one GPU: the count of the current device, once)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    const int total = N * tiles, per_block = cdiv(total, cus), blocks = cdiv(total, per_block);
    ALIVE_CHECK_ARG(ws != nullptr && ws_bytes >= (int64_t)blocks * NCONV * G::GUARD, "%s: workspace too small (see the _workspace_bytes query)", name);
    // ALIVE_FB256_WAVES = 4: one wave per SIMD (64 channels x 128 columns each); default: two (32 channels x 128 columns each)
    static const bool two = !(getenv("ALIVE_FB256_WAVES") && atoi(getenv("ALIVE_FB256_WAVES")) == 4);
    if (two)
        filter_block256_kernel<C, 1><<<blocks, Geo<C, 1>::NT, G::LDS_BYTES, (hipStream_t)stream>>>(U, L, wts, film, film_rows, Lf, film_off, ratio, t0, f0,
                                                                                                   film_ld, skip, out, tiles, per_block, total, (unsigned char*)ws);
    else
        filter_block256_kernel<C, 2><<<blocks, Geo<C, 2>::NT, G::LDS_BYTES, (hipStream_t)stream>>>(U, L, wts, film, film_rows, Lf, film_off, ratio, t0, f0,
                                                                                                   film_ld, skip, out, tiles, per_block, total, (unsigned char*)ws);
    ALIVE_CHECK_LAUNCH(name);
#ifdef ALIVE_FB256_PROF
    {
        unsigned long long v[16], z[16] = {0};
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(fb256_prof), sizeof(v));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(fb256_prof), z, sizeof(z));
        fprintf(stderr, "%s cycles: start->loads issued %llu | barrier %llu | z0 %llu | ctx+B1 %llu | k-loop %llu | B2 %llu | epilogue+B3 %llu | store %llu\n",
                name, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    }
#endif
    return ALIVE_OK;
}
}  // namespace

// U[N][C][L] (the block's residual stream: the output of the composed transposed conv) -> out = FilterBlock(U) + skip, fp32.
// w16[q] / bias[q], q = 0 .. 5: blocks[q / 2].c1 / .c2 -- the fp16 slab of module/_pack.py::pack_conv_split_h and the fp32 bias.
// film[N][film_rows][film_ld]: rows film_off + q * 2 C + (0 .. C - 1 scale | C .. 2 C - 1 shift) for conv q; frame range as alive_filter_block64_range.
// workspace: six contexts (16 columns x C channels, fp16) per block, at most one block per tile (C = 64: the contexts stay in LDS and the
// workspace is not touched; the argument is checked all the same)
extern "C" int64_t alive_filter_block256_workspace_bytes(int N, int L) {
    return N > 0 && L > 0 ? (int64_t)N * cdiv(L, Geo<256, 1>::BL) * NCONV * Geo<256, 1>::GUARD : 0;
}
extern "C" int alive_filter_block256_fp16(const float* U, int N, int L, const void* const* w16, const float* const* bias, const float* film,
                                          int film_rows, int Lf, int film_off, int t0, int f0, int film_ld, const float* skip, float* out,
                                          void* ws, int64_t ws_bytes, void* stream) {
    return fb_launch<256>("alive_filter_block256_fp16", U, N, L, w16, bias, film, film_rows, Lf, film_off, t0, f0, film_ld, skip, out, ws, ws_bytes, stream);
}
extern "C" int64_t alive_filter_block64s_workspace_bytes(int N, int L) {
    return N > 0 && L > 0 ? (int64_t)N * cdiv(L, Geo<64, 1>::BL) * NCONV * Geo<64, 1>::GUARD : 0;
}
extern "C" int alive_filter_block64s_fp16(const float* U, int N, int L, const void* const* w16, const float* const* bias, const float* film,
                                          int film_rows, int Lf, int film_off, int t0, int f0, int film_ld, const float* skip, float* out,
                                          void* ws, int64_t ws_bytes, void* stream) {
    return fb_launch<64>("alive_filter_block64s_fp16", U, N, L, w16, bias, film, film_rows, Lf, film_off, t0, f0, film_ld, skip, out, ws, ws_bytes, stream);
}

ALIVE_F16_SAT_GETTER(alive_f16_sat_filter_big)
