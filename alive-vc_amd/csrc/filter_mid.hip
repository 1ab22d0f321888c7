// FilterBlock.forward for the 64-channel scale of the decoder's U-Net (/root/reference/module/decoder.py:105-150),
// fused into ONE kernel like the 16- / 8-channel scales (filter_small.hip), on the split-bf16 MFMA.
//
// Conv by conv this scale (36 000 samples x 64 channels per window) is HBM-bound: every k5 conv reads and writes one or
// two full fp32 tensors (4.7 GB per conv at 128 windows against 0.5 ms of MFMA work).  Here a block owns a tile of
// 256 columns (200 outputs + the 56-column causal halo 4 x (1+1+2+2+4+4), recomputed instead of exchanged) and keeps
// it on chip through the 1x1 input conv and the six GELU -> FiLM -> reflect-left causal k5 convs:
//   * the modulated conv input z and the intermediate y live in two LDS buffers, ALREADY SPLIT into two bf16 planes
//     (hi = bf16(v), lo = bf16(v - hi)) and stored [column][64 channels] so that an MFMA B fragment is one ds_read_b128;
//   * a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulate (the "bf16x3" product of conv_split.hip, ~2^-16 per product).
// HBM traffic drops to the input, the U-Net skip and the output (3 tensor passes instead of ~18).
//
// Round 4: rewritten for v_mfma_f32_32x32x16_bf16 as ONE software pipeline over (conv, column tile) steps.
// What rounds 1-3's kernel (four waves, 16x16x32, 7.3 ms per 128 windows) was bound by, measured with in-kernel stamps,
// ablation builds and a microbenchmark of instruction costs beside MFMAs (tools/stamp_fb64.py, tools/mfma_filler.py,
// DESIGN.md 3.2b'): nothing overlapped.  A block's time was the SUM of its MFMA time, its epilogue time, its LDS fragment
// reads and its exposed memory latencies, and an eight-wave version (two waves per SIMD) ran the same 7.3 ms: one SIMD
// issues ONE vector instruction stream, whichever wave it comes from, and beside an MFMA an instruction costs what it costs --
// MFMA 8 cycles of issue, plain VALU 4.4, v_exp / v_rcp 8, v_accvgpr_read 8, dependent VALU 8 (latency), ds_read_b128 16 and
// ds_write_b64 24 per wave with four waves on the LDS; only ~24 cycles' worth hide in the 32 cycles of a 32x32x16 MFMA, and a
// 16x16x32 MFMA (8 of 16 cycles) hides nothing.  So:
//   * 32x32x16: half the MFMA issue cost and half the fragment reads per FLOP.  A wave owns 32 output channels (rg = w & 1) and
//     four column tiles of 32 -- tiles chalf, chalf + 2, chalf + 4, chalf + 6 of the block's eight (chalf = w >> 1); its
//     A fragments (20 k-steps x 2 planes = 160 registers) stay in registers for a conv, the residual stream (4 x 16) too: one
//     wave per SIMD, the whole 512-register file.
//   * every step runs the 60 MFMAs of one column tile and, cut into stages of ~16 INDEPENDENT vector instructions per k-step
//     (four elements = four dependency chains per stage, one scheduling region per k-step -- hipcc otherwise puts 14 MFMAs back
//     to back and the whole job behind them), the epilogue of the PREVIOUS step's tile: GELU -> FiLM -> re-split -> LDS.
//   * the interleaved tile ownership is what lets the pipeline run across conv boundaries: column tile t of conv q + 1 needs
//     tiles t - 1 and t of conv q, i.e. the k-th tile of a wave needs k-th and (k-1)-th tiles of the others, never their last
//     one -- so the last tile's epilogue of conv q runs under the first tile's MFMAs of conv q + 1 (one barrier per step instead
//     of a pipeline fill and drain per conv), and a conv's weights are reloaded k-step by k-step behind its last tile's MFMAs.
//   * LDS image [column][64 channels] bf16, hi and lo plane, 16-byte chunks XOR-swizzled by (row >> 1) & 7: a 32x32x16 B fragment
//     is read by 32 lanes of consecutive rows and the ds_read_b128 lane groups hold eight even and eight odd rows each, which
//     (row >> 1) & 7 spreads over all eight chunk positions at any tap shift.  FiLM rows as [conv][channel][9 frames] (scale,
//     shift) pairs: one ds_read2_b64 per element fetches both frames.
//   * FIRST = the first tile of a window (ReflectionPad1d at t = 0, common.py:88) is an instantiation of its own with per-lane
//     fragment addresses; every other tile reads its fragments at lane-constant bases + immediates.  Its leftmost column tile
//     reaches up to 16 rows left of the LDS image: a 2-KB guard in front of it makes those reads legal, and what they return
//     only reaches halo columns (the halo IS the dependency cone of the six convs).
// 7.3 -> see DESIGN.md for the measured time.
//
// A hazard found on the way (and the likely mechanism of the run-to-run defect of the 16x16x32 kernel, DESIGN.md 3.2b'): when the
// first MFMA of a NEW accumulation chain follows the last MFMA of the previous chain within ~12 wait states and the previous
// chain's accumulator is read (v_accvgpr_read) only later, beside the new chain's MFMAs, the LAST register of that accumulator
// comes back wrong -- deterministically for some tiles, run to run for others.  hipcc 7.2 inserts the documented wait states for
// the first read only.  Eight more wait states between the chains (s_nop below), or reading the accumulator at once, make every
// launch exact (tools/stress_filter_block.py: 0 of 100 differ; -DALIVE_FB64_NO_CHAIN_GAP reproduces the failure).
#include "conv_epilogue.h"
#include <type_traits>
#include <stdlib.h>

// The Makefile compiles this file with -fno-slp-vectorize and says so with the macro: packed fp32 math beside MFMAs is an
// anti-lever on this part (a v_pk_fma_f32 costs ~22 cycles more than two v_fma_f32 there, MI355X_MICROARCH.md), and the SLP
// vectorizer is also what produced the non-deterministic binary of the 16x16x32 kernel.
#ifndef ALIVE_FILTER_MID_NO_SLP
#error "filter_mid.hip must be compiled with -fno-slp-vectorize -DALIVE_FILTER_MID_NO_SLP (see csrc/Makefile)"
#endif

static long long* g_stamps64 = nullptr;

namespace {

constexpr int C = 64;
constexpr int HALO = 56;
constexpr int NCONV = 6;
constexpr int BL = 256;                 // columns per tile incl. halo
constexpr int TT = BL - HALO;           // 200 output columns per tile
constexpr int NFP = 8;                  // FiLM frames a tile may span
constexpr int NFS = NFP + 1;            // staged frames per channel (i0 <= NFP - 1, i1 = i0 + 1)
constexpr int ROWB = 128;               // bytes per LDS row (64 channels bf16)
constexpr int PLANE = BL * ROWB;        // 32 KB
constexpr int BUF = 2 * PLANE;          // hi + lo
constexpr int W_IN = 2 * C * C;         // bf16 elements of the input conv [2][64][64]
constexpr int W_K5 = 2 * C * 5 * C;     // bf16 elements of a k5 conv [2][64][320]
constexpr int W_K5H = C * 5 * C;        // fp16 elements of a k5 conv [64][320]: the slab behind the bf16 packs (H16 below)
constexpr int NT = 4;                   // column tiles of 32 per wave
constexpr int TROW = 32 * ROWB;         // bytes per column tile in a plane
constexpr int TSTEP = 2 * TROW;         // a wave's consecutive tiles are two column tiles apart
constexpr int GUARD = 16 * ROWB;        // 2 KB in front of the plane buffers
constexpr int LDS_BYTES = GUARD + 2 * BUF + NCONV * C * NFS * 8 + BL * 8;     // 162816 B

__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ int swz(int r) { return (r >> 1) & 7; }

// NTT = column tiles of 32 per wave: 4 on the batch path (256-column tiles, 200 outputs), 2 for a signal of a few such tiles (the
// streaming step: 800 samples = four batch tiles on four CUs, 72 us of a 0.8-ms step): 128-column tiles, 72 outputs, twelve blocks with
// half the serial chain each.  The halo is then 44 % of a tile -- irrelevant where the chip is empty.
// H16 (round 5, decoder precision mode 1 with ALIVE_DECODER_BF16_MASK bit 16): the six k5 convs multiply ONE fp16 plane of the modulated
// tensor by ONE fp16 plane of the weights -- a third of the MFMAs, no lo plane in LDS, half the weight registers.  The 1x1 input conv
// keeps the split form (its operand is the raw residual stream).
// SWEEP (round 6; H16 only, batch path): a block walks a SEGMENT of a window left to right, 256 NEW columns per tile, and hands every
// conv's causal context -- the last 16 columns of that conv's input, 2 KB -- to the next tile through LDS instead of recomputing a
// 56-column halo per 200 outputs (22 % of the kernel's work).  A segment starts with one warm-up tile whose output is not stored (its
// own first 56 columns are computed from nothing; the context it leaves is clean).  Same columns from the same operands in the same order:
// bit for bit the tiled form's output.  c_first: first column of segment 0 (the window's first 200 columns stay with the FIRST form,
// which reflects at t = 0); seg_cols: columns per segment, a multiple of 256.
template <bool FIRST, int NTT = 4, bool H16 = false, bool SWEEP = false>
__global__ __launch_bounds__(256, 1) void filter_block64_kernel(const float* __restrict__ U, int L,
                                                                const unsigned short* __restrict__ W16,
                                                                const float* __restrict__ biases,
                                                                const float* __restrict__ film, int film_rows, int Lf,
                                                                int film_off, float ratio, int t_off, int f_off, int film_ld,
                                                                const float* __restrict__ skip, float* __restrict__ out, long long* stamps,
                                                                int c_first = 0, int seg_cols = 0) {
    static_assert(!SWEEP || (H16 && !FIRST && NTT == 4), "the sweep form exists for the fp16 batch tiles only (it parks the context in the unused lo plane)");
#ifdef ALIVE_STAMPS                 // diagnostic build only (tools/ab_build.sh x.so filter_mid.hip -DALIVE_STAMPS; tools/stamp_fb64.py)
    long long ts3[12];
    int nts = 0;
#define STAMP3() ts3[nts++] = wall_clock64()
#else
#define STAMP3()
#endif
    STAMP3();
    constexpr int HALO_ = SWEEP ? 0 : HALO;            // columns of a tile that are recomputed context
    constexpr int NT = NTT, BL = 64 * NT, TT = BL - HALO_, PLANE = BL * ROWB, BUF = 2 * PLANE;      // (shadow the batch constants above)
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* bufZ = sm + GUARD;
    unsigned char* bufY = bufZ + BUF;
    f32x2* Fs = (f32x2*)(bufZ + 2 * BUF);             // [NCONV][C][NFS] (scale, shift)
    uint2* Xc = (uint2*)(Fs + NCONV * C * NFS);       // [BL] per column: (8 * i0 relative to the staged frames, w1)

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = w & 1, chalf = w >> 1;
    const int n32 = lane & 31, lh = lane >> 5;
    const int n = blockIdx.y;
    const float* Un = U + (size_t)n * C * L;
    // sweep form: this block's segment, the context slots (one per conv, in the lo plane of bufY, which the fp16 form never touches) and
    // the 16 rows in front of either buffer that a tile's leftmost fragments read (bufY's: the tail of bufZ's lo plane, free once the
    // input conv has read the raw tile)
    const int seg_begin = SWEEP ? c_first + (int)blockIdx.x * seg_cols : 0;
    const int seg_end = SWEEP ? (seg_begin + seg_cols < L ? seg_begin + seg_cols : L) : 0;
    unsigned char* const hist = bufY + PLANE;
    unsigned char* const guardZ = bufZ - GUARD;
    unsigned char* const guardY = bufY - GUARD;
    auto copy2k = [&](unsigned char* dst, const unsigned char* src) {          // one wave: 2 KB = 16 rows of 128 B, swizzle preserved (256 = 0 mod 16)
        *(u32x4*)(dst + lane * 16) = *(const u32x4*)(src + lane * 16);
        *(u32x4*)(dst + 1024 + lane * 16) = *(const u32x4*)(src + 1024 + lane * 16);
    };
    if constexpr (SWEEP) {
        if (seg_begin >= L) return;
        for (int i = tid; i < (NCONV * GUARD) / 16; i += 256) ((u32x4*)hist)[i] = u32x4{0u, 0u, 0u, 0u};
        for (int i = tid; i < GUARD / 16; i += 256) ((u32x4*)guardZ)[i] = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }
#pragma unroll 1
    for (int it = SWEEP ? -1 : 0;; ++it) {
    const bool warm = SWEEP && it < 0;                 // the segment's warm-up tile: computed for its context, not stored
    const float* biases_i = biases;
    const unsigned short* W16_i = W16;
    int n32 = lane & 31, lh = lane >> 5, rg = w & 1, chalf = w >> 1;        // (shadow the kernel-scope values: opaque per tile in the sweep form)
    if constexpr (SWEEP) {
        asm volatile("" : "+s"(biases_i));
        asm volatile("" : "+s"(W16_i));
        asm volatile("" : "+v"(n32), "+v"(lh));
        asm volatile("" : "+s"(rg), "+s"(chalf));
    }
    const int t0 = SWEEP ? seg_begin + it * BL : (FIRST ? blockIdx.x * TT : (blockIdx.x + 1) * TT);      // FIRST: the host launches it for tile 0 (or for every tile of a small problem)
    if (SWEEP && it >= 0 && t0 >= seg_end) break;
    const int tbase = t0 - HALO_;

    // ---- stage the raw input tile (split into planes; thread = column) and the FiLM rows of the tile: every global load of the
    //      prologue is issued before the first LDS write, so one memory latency covers all of them ----
    const int f_lo = lerp_coord((tbase < 0 ? 0 : tbase) + t_off, ratio, Lf).i0;     // frames of the WINDOW (t_off: range mode)
    {
        // thread = (column, part): NPT = 256 / BL threads share a column, each takes C / NPT channels (batch tiles: one thread per column)
        constexpr int NPT = 256 / BL, CPT = C / NPT;
        const int col = tid % BL, part = tid / BL;
        const int t = tbase + col;
        const bool ok = t >= 0 && t < L;
        const float* uc = Un + (ok ? t : 0) + (size_t)(part * CPT) * L;
        float v[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) v[c] = uc[(size_t)c * L];
        constexpr int NFL = NCONV * 2 * C * NFS / 256;        // 27 FiLM values per thread
        float fv[NFL];
#pragma unroll
        for (int k = 0; k < NFL; ++k) {
            const int e = tid + 256 * k;
            const int f = e % NFS, c = (e / NFS) % C, sel = (e / (NFS * C)) & 1, q = e / (NFS * C * 2);
            int fr = f_lo + f;
            fr = fr < Lf ? fr : Lf - 1;
            int fc = fr - f_off;                             // frame of the window -> column of the film tensor
            fc = fc < 0 ? 0 : (fc < film_ld ? fc : film_ld - 1);
            fv[k] = film[((size_t)n * film_rows + film_off + q * 2 * C + sel * C + c) * film_ld + fc];
        }
        if (part == 0) {   // F.interpolate coordinates of this thread's column, relative to the staged FiLM frames (i1 = i0 + 1: the table
            // repeats the window's last frame, which is what the clamp of upsample_linear1d reads there)
            const int tc = t < 0 ? 0 : (t < L ? t : L - 1);
            const Lerp lp = lerp_coord(tc + t_off, ratio, Lf);
            int i0 = lp.i0 - f_lo;
            i0 = i0 < NFP - 1 ? i0 : NFP - 1;
            Xc[col] = make_uint2((unsigned)(i0 * 8), __float_as_uint(lp.w1));
        }
#pragma unroll
        for (int ck = 0; ck < CPT / 8; ++ck) {
            u32x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = ok ? v[ck * 8 + 2 * e] : 0.0f, x1 = ok ? v[ck * 8 + 2 * e + 1] : 0.0f;
                const unsigned h = pack2(x0, x1);
                hi[e] = h;
                lo[e] = pack2(x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xffff0000u));
            }
            unsigned char* dst = bufZ + col * ROWB + (((part * (CPT / 8) + ck) ^ swz(col)) << 4);
            *(u32x4*)dst = hi;
            *(u32x4*)(dst + PLANE) = lo;
        }
#pragma unroll
        for (int k = 0; k < NFL; ++k) {
            const int e = tid + 256 * k;
            const int f = e % NFS, c = (e / NFS) % C, sel = (e / (NFS * C)) & 1, q = e / (NFS * C * 2);
            // the scale rows are stored halved (exact): gelu(x) * sc + sh = (x + |x| erf|x / sqrt 2|) * (sc / 2) + sh saves the 0.5 x
            ((float*)Fs)[((q * C + c) * NFS + f) * 2 + sel] = sel == 0 ? 0.5f * fv[k] : fv[k];
        }
        if constexpr (SWEEP) {
            if (w == 0) copy2k(guardZ, hist);                    // conv 0's context: the last 16 columns of the previous tile's z0
        }
    }
    __syncthreads();
    STAMP3();

    // residual stream: h[i][4 g + e] = channel 32 rg + 8 g + 4 lh + e of column 32 (2 i + chalf) + n32 (MFMA C layout)
    f32x16 h[NT];

    // lane-constant pieces of the per-tile addresses: tile i of this wave adds the immediate i * TSTEP
    int colw = chalf * 32 + n32;                                                   // column of tile i: colw + 64 i
    if constexpr (SWEEP) asm volatile("" : "+v"(colw));       // (a loop body now: without the opaque uses below hipcc hoists every lane-constant
                                                              //  address, bias tuple and weight pointer of all seven stages out of the tile loop
                                                              //  -- 512 registers and 132 spilled)
    int st_off[4];                                                                 // modulated planes: channels 32 rg + 8 g + 4 lh .. +3
#pragma unroll
    for (int g = 0; g < 4; ++g) st_off[g] = colw * ROWB + (((4 * rg + g) ^ swz(colw)) << 4) + 8 * lh;
    const uint2* xcw = Xc + colw;
    const unsigned char* fsw = (const unsigned char*)(Fs + (32 * rg + 4 * lh) * NFS);

    // FiLM operands of one channel group (4 channels) of a column: (scale, shift) at frames i0 and i0 + 1
    struct FilmG {
        f32x2 a0[4], a1[4];
    };
    // The epilogue of one column tile (an "item"): v = chain result (+ residual, c2) -> GELU -> FiLM of the next conv's input ->
    // re-split -> LDS planes, per channel group g in five stages of four independent chains each
    struct Epi {
        int off;
        float w0, w1;
        FilmG F[2];
        float z[4], jt[4], je[4], jsc[4], jsh[4], jx[4];
    };
    auto film_fetch = [&](int qf, int off, int g, FilmG& F) {
        const unsigned char* f = fsw + qf * (C * NFS * 8) + off;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2* fp = (const f32x2*)(f + ((8 * g + e) * NFS) * 8);
            F.a0[e] = fp[0];
            F.a1[e] = fp[1];
        }
    };
    auto epi_begin = [&](Epi& E, int qf, int t) {
        const uint2 xc = xcw[64 * t];
        E.off = (int)xc.x;
        E.w1 = __uint_as_float(xc.y);
        E.w0 = 1.0f - E.w1;
        film_fetch(qf, E.off, 0, E.F[0]);
    };
    // second: the item is a c2 result (v = chain + residual stream, which it replaces); emit: a next conv exists (FiLM rows qf)
    // and takes the modulated planes at dstp
    auto epi_stage = [&](Epi& E, bool second, bool emit, int qf, unsigned char* dstp, int t, const f32x16& accv, int g, int st) {
        const FilmG& Fg = E.F[g & 1];
        if (st == 0) {
            if (emit && g + 1 < 4) film_fetch(qf, E.off, g + 1, E.F[(g + 1) & 1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = accv[4 * g + e];                 // one v_accvgpr_read per element (8 cycles beside an MFMA): kept for stage 3
                if (second) {
                    x = x + h[t][4 * g + e];
                    h[t][4 * g + e] = x;
                }
                E.jx[e] = x;
                E.jt[e] = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.3275911f * 0.70710678118654752440f, 1.0f));
                const float xs = x * 0.84932180028801904272f;         // exp(-x^2 / 2) = exp2(-(x sqrt(log2(e) / 2))^2)
                E.je[e] = __builtin_amdgcn_exp2f(-(xs * xs));
            }
            // (the empty asm pins a stage's results to its own scheduling region: pure arithmetic otherwise sinks to its consumer's)
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(E.jt[e]), "+v"(E.je[e]), "+v"(E.jx[e]));
        } else if (!emit) {
        } else if (st == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                E.jsc[e] = fmaf(E.w0, Fg.a0[e][0], E.w1 * Fg.a1[e][0]);        // fma(w0, a, round(w1 * b)): ATen's linear interp
                E.jsh[e] = fmaf(E.w0, Fg.a0[e][1], E.w1 * Fg.a1[e][1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(E.jsc[e]), "+v"(E.jsh[e]));
        } else if (st == 2) {          // erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7), like gelu_fast
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float p = fmaf(1.061405429f, E.jt[e], -1.453152027f);
                p = fmaf(p, E.jt[e], 1.421413741f);
                p = fmaf(p, E.jt[e], -0.284496736f);
                p = fmaf(p, E.jt[e], 0.254829592f);
                E.jt[e] = p * E.jt[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(E.jt[e]));
        } else if (st == 3) {          // x (1 + sign(x) erf|x / sqrt 2|) = x + |x| erf_abs = 2 gelu(x); jsc is half the FiLM scale
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float erf_abs = fmaf(-E.jt[e], E.je[e], 1.0f);
                E.z[e] = fmaf(fmaf(fabsf(E.jx[e]), erf_abs, E.jx[e]), E.jsc[e], E.jsh[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(E.z[e]));
        } else {
            const float* z = E.z;
            unsigned char* p = dstp + st_off[g] + t * TSTEP;
            if constexpr (H16) {
                // one fp16 plane, saturating; saturations are counted in the tile's OUTPUT columns only: the halo columns are its
                // neighbour's outputs, and left of their dependency cone they hold arbitrary values
                const bool cnt = SWEEP ? !warm : colw + 64 * t >= HALO;
                *(uint2*)p = make_uint2(pack_f16x2(z[0], z[1], cnt), pack_f16x2(z[2], z[3], cnt));
            } else {
                const unsigned h01 = pack2(z[0], z[1]), h23 = pack2(z[2], z[3]);
                const unsigned l01 = pack2(z[0] - __uint_as_float(h01 << 16), z[1] - __uint_as_float(h01 & 0xffff0000u));
                const unsigned l23 = pack2(z[2] - __uint_as_float(h23 << 16), z[3] - __uint_as_float(h23 & 0xffff0000u));
                *(uint2*)p = make_uint2(h01, h23);
                *(uint2*)(p + PLANE) = make_uint2(l01, l23);
            }
        }
    };

    // weights of the k5 convs: k-step s = 4 j + cb (tap j, 16-channel block cb), k = j * 64 + cb * 16 + 8 lh
    bf16x8 a[20][2];
    auto load_weights_step = [&](int q, int s) {
        if constexpr (H16) {
            const unsigned short* Wq = W16_i + W_IN + (size_t)NCONV * W_K5 + (size_t)q * W_K5H + (size_t)(32 * rg + n32) * (5 * C) + 8 * lh;
            a[s][0] = *(const bf16x8*)(Wq + s * 16);
        } else {
            const unsigned short* Wq = W16_i + W_IN + (size_t)q * W_K5 + (size_t)(32 * rg + n32) * (5 * C) + 8 * lh;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[s][pl] = *(const bf16x8*)(Wq + (size_t)pl * C * 5 * C + s * 16);
        }
    };

    // ---- input_conv (1x1, K = 64 = 4 k-steps): h = Win * U + b ----
    {
        bf16x8 ai[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                ai[s][pl] = *(const bf16x8*)(W16_i + (size_t)pl * C * C + (size_t)(32 * rg + n32) * C + s * 16 + 8 * lh);
        f32x16 b16;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) b16[4 * g + e] = biases_i[32 * rg + 8 * g + 4 * lh + e];
        const int bw = colw * ROWB + ((lh ^ swz(colw)) << 4);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#ifndef ALIVE_FB64_NO_CHAIN_GAP
            if (i > 0) ALIVE_CHAIN_GAP(7);                    // h[i - 1] stays in its accumulator registers and is read by the epilogues below
#endif
            f32x16 acc = b16;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned char* bp = bufZ + ((bw ^ (s << 5)) + i * TSTEP);
                const bf16x8 bh = *(const bf16x8*)bp, bl = *(const bf16x8*)(bp + PLANE);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai[s][1], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai[s][0], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai[s][0], bh, acc, 0, 0, 0);
            }
            h[i] = acc;
        }
    }
    __syncthreads();                       // every wave has read the raw columns
#pragma unroll
    for (int s = 0; s < 20; ++s) load_weights_step(0, s);        // 40 loads per lane in flight under the epilogues below
    // z0 = mod_0(h) over the raw columns: the first three tiles here, the fourth under the first MFMAs of conv 0 as a "c2 item"
    // whose chain result is zero (v = 0 + h)
    f32x16 acc[2];
    {
        f32x16 zero16;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
        acc[1] = zero16;
#pragma unroll
        for (int i = 0; i < NT - 1; ++i) {
            Epi E;
            epi_begin(E, 0, i);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int st = 0; st < 5; ++st) epi_stage(E, true, true, 0, bufZ, i, zero16, g, st);
        }
    }
    __syncthreads();
    STAMP3();

    // ---- three FilterResBlocks: q = 2 j (c1), 2 j + 1 (c2), dilation 2^j    (decoder.py:128-134) ----
    // One conv = four steps; step i runs the MFMAs of the wave's tile i (-> acc[i & 1]) and the epilogue of the previous step's
    // tile (tile i - 1 of this conv, or tile 3 of the previous one), then a barrier.  SECOND: this conv is the c2 of its block.
    auto conv = [&](auto second_tag, const int q) {
        constexpr bool SECOND = decltype(second_tag)::value;
        const unsigned char* in = SECOND ? bufY : bufZ;
        unsigned char* dst = SECOND ? bufZ : bufY;
        const bool emit = q + 1 < NCONV;
        f32x16 b16;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) b16[4 * g + e] = biases_i[(1 + q) * C + 32 * rg + 8 * g + 4 * lh + e];
        const int d = 1 << (q >> 1);
        // B fragment of k-step s = 4 j + cb of tile i: row r = colw + 64 i + (j - 4) d, 16 B at r * 128 + ((2 cb + lh) ^ swz(r)) * 16
        // = base[s] + i * TSTEP (swz does not depend on i): twenty lane-constant registers per conv, immediates per tile -- an XOR
        // per fragment read in the loop is 0.7 vector instructions per MFMA where only five are free
        int base[20];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int r0 = colw + (j - 4) * d;                 // >= -16: the guard
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) base[4 * j + cb] = (r0 * ROWB + ((lh ^ swz(r0)) << 4)) ^ (cb << 5);
        }
        auto frag_off = [&](int i, int s) -> int {
            if (FIRST) {                                        // reflect at t = 0 (column HALO of this tile), per lane
                int ta = tbase + colw + 64 * i + ((s >> 2) - 4) * d;
                ta = ta < 0 ? -ta : ta;
                int r = ta - tbase;
                r = r < BL ? r : BL - 1;
                return (r * ROWB + ((lh ^ swz(r)) << 4)) ^ ((s & 3) << 5);
            }
            return base[s] + i * TSTEP;
        };
        constexpr int PF = 3, NSLOT = PF + 1;                   // fragments are requested PF k-steps ahead of their MFMAs
        bf16x8 fh[NSLOT], fl[NSLOT];
        auto frag_load = [&](int i, int s, int slot) {
            const unsigned char* bp = in + frag_off(i, s);
            fh[slot] = *(const bf16x8*)bp;
            if constexpr (!H16) fl[slot] = *(const bf16x8*)(bp + PLANE);
        };
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            // the item of this step: tile i - 1 of this conv, or (i == 0) tile 3 of the previous conv -- whose type is the other one,
            // which always has a next conv, whose FiLM rows are this conv's and whose planes are this conv's input
            const bool it_second = i > 0 ? SECOND : !SECOND;
            const bool it_emit = i > 0 ? emit : true;
            const int it_qf = i > 0 ? q + 1 : q;
            unsigned char* it_dst = i > 0 ? dst : (unsigned char*)in;
            const int it_t = i > 0 ? i - 1 : NT - 1;
            const f32x16& it_acc = acc[(i + 1) & 1];
#ifndef ALIVE_FB64_NO_CHAIN_GAP
            ALIVE_CHAIN_GAP(15);                              // see the header: distance between two accumulation chains
#endif
            if constexpr (SWEEP) {
                // this conv's input is complete since the barrier of step 0 (its last tile was the pending item): keep its last 16
                // columns for the next tile; one step later hand the NEXT conv its context (it reads the other buffer, whose front
                // rows nobody reads before that conv starts)
                if (i == 1 && w == 0) copy2k(hist + q * GUARD, in + (BL - 16) * ROWB);
                if (i == 2 && w == 0 && q + 1 < NCONV) copy2k(SECOND ? guardZ : guardY, hist + (q + 1) * GUARD);
            }
            Epi E;
            if (it_emit) epi_begin(E, it_qf, it_t);
            // Two tiles per wave, window start: the reflected fragments of tile 0 (columns up to HALO + 47) lie in the columns of the
            // pending item (tile 1 of the conv before: 64 ..), which this step would only be writing -- that block finishes the item
            // first.  (Four tiles per wave: the pending item is columns 192 .., nothing reflects that far.)
            const bool drain_first = NT == 2 && FIRST && tbase < 0 && i == 0;          // block-uniform
            if (drain_first) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int st = 0; st < 5; ++st) epi_stage(E, it_second, it_emit, it_qf, it_dst, it_t, it_acc, g, st);
                __syncthreads();
            }
            acc[i & 1] = b16;
#pragma unroll
            for (int s = 0; s < PF; ++s) frag_load(i, s, s);
#pragma unroll
            for (int s = 0; s < 20; ++s) {
                if (s + PF < 20) frag_load(i, s + PF, (s + PF) % NSLOT);
                if constexpr (H16) {
                    acc[i & 1] = mfma_f16(a[s][0], fh[s % NSLOT], acc[i & 1]);
                } else {
                    acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], fh[s % NSLOT], acc[i & 1], 0, 0, 0);
                    acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], fl[s % NSLOT], acc[i & 1], 0, 0, 0);
                    acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], fh[s % NSLOT], acc[i & 1], 0, 0, 0);
                }
                if (i == NT - 1 && emit) load_weights_step(q + 1, s);     // a[s] is dead: the next conv's weights travel under the rest of the step
                if (!drain_first) epi_stage(E, it_second, it_emit, it_qf, it_dst, it_t, it_acc, s / 5, s % 5);
#ifdef ALIVE_FB64_DUP_STAGE           // diagnostic (timing only): one stage's work a second time -- the increment prices the stage
                if (s % 5 == ALIVE_FB64_DUP_STAGE && it_emit) {
                    Epi E2 = E;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { asm volatile("" : "+v"(E2.jt[e]), "+v"(E2.je[e]), "+v"(E2.jx[e]), "+v"(E2.jsc[e])); asm volatile("" : "+v"(E2.jsh[e]), "+v"(E2.z[e])); }
                    epi_stage(E2, false, true, it_qf, it_dst, it_t, it_acc, s / 5, s % 5);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { asm volatile("" :: "v"(E2.jt[e]), "v"(E2.je[e]), "v"(E2.jx[e]), "v"(E2.jsc[e])); asm volatile("" :: "v"(E2.jsh[e]), "v"(E2.z[e])); }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);           // one scheduling region per k-step
            }
            __syncthreads();
        }
    };
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
        conv(std::false_type{}, 2 * j);
        STAMP3();
        conv(std::true_type{}, 2 * j + 1);
        STAMP3();
    }
    // drain: the last conv's last tile only adds into the residual stream
#pragma unroll
    for (int r = 0; r < 16; ++r) h[NT - 1][r] = acc[1][r] + h[NT - 1][r];

    // ---- store the tile (+ U-Net skip, decoder.py:191): through LDS so that global accesses are 16-B vectors along t ----
    float* Ht = (float*)bufZ;                         // [64][BL + 4] fp32 = 66.5 KB over bufZ / bufY
    constexpr int HP = BL + 4;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) Ht[(32 * rg + 8 * g + 4 * lh + e) * HP + colw + 64 * i] = h[i][4 * g + e];
    __syncthreads();
    constexpr int NV = (C * (TT / 4) + 255) / 256;          // 13 (sweep: 16) vectors of 4 columns per thread
    if (!warm) {                                            // (block-uniform)
    f32x4 sk[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int g = tid + 256 * i;
        const int co = g / (TT / 4), c4 = (g - co * (TT / 4)) * 4;
        const int t = t0 + c4;
        sk[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (skip != nullptr && g < C * (TT / 4) && t + 3 < L) sk[i] = *(const f32x4*)(skip + ((size_t)n * C + co) * L + t);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int g = tid + 256 * i;
        const int co = g / (TT / 4), c4 = (g - co * (TT / 4)) * 4;
        const int t = t0 + c4;
        if (g >= C * (TT / 4) || t >= L) continue;
        const size_t o = ((size_t)n * C + co) * L + t;
        const f32x4 v = *(const f32x4*)&Ht[co * HP + HALO_ + c4];
        if (t + 3 < L) {
            *(f32x4*)(out + o) = v + sk[i];
        } else {
            for (int e = 0; e < 4 && t + e < L; ++e) out[o + e] = v[e] + (skip != nullptr ? skip[o + e] : 0.0f);
        }
    }
    }
    if (!SWEEP) break;
    __syncthreads();                                        // the next tile's staging overwrites the buffers Ht lies over
    }      // tiles of the segment
#ifdef ALIVE_STAMPS
    // per block 32 words: wave 0 -> [0..8] phase durations (staging, input conv, six convs, store); wave 2 -> [16..]
    if (stamps != nullptr && (tid == 0 || tid == 128)) {
        long long* o = stamps + ((size_t)blockIdx.y * (gridDim.x + 1) + (FIRST ? blockIdx.x : blockIdx.x + 1)) * 32 + (tid ? 16 : 0);
        for (int i = 0; i + 1 < nts; ++i) o[i] = ts3[i + 1] - ts3[i];
        o[nts - 1] = wall_clock64() - ts3[nts - 1];
    }
#endif
}

}  // namespace

extern "C" void alive_debug_set_stamps64(long long* p) { g_stamps64 = p; }
extern "C" int64_t alive_filter_block64_weights(void) { return (int64_t)W_IN + (int64_t)NCONV * W_K5 + (int64_t)NCONV * W_K5H; }

extern "C" int alive_filter_block64(const float* U, int N, int L, const void* W16, const float* biases, const float* film,
                                    int film_rows, int Lf, int film_off, const float* skip, float* out, void* stream) {
    return alive_filter_block64_range(U, N, L, W16, biases, film, film_rows, Lf, film_off, 0, 0, Lf, skip, out, stream);
}

namespace {
template <bool H16>
int filter_block64_impl(const float* U, int N, int L, const void* W16, const float* biases, const float* film,
                        int film_rows, int Lf, int film_off, int t0, int f0, int film_ld, const float* skip,
                        float* out, void* stream) {
    ALIVE_CHECK_ARG(U && W16 && biases && film && out, "alive_filter_block64: null pointer");
    ALIVE_CHECK_ARG(N > 0 && L > 16 && Lf > 0, "alive_filter_block64: bad sizes (L must exceed the largest reflect pad, 16)");
    ALIVE_CHECK_ARG(U != out, "alive_filter_block64: in-place not supported (tiles read a halo of their left neighbour)");
    ALIVE_CHECK_ARG((L & 3) == 0 && ((((uintptr_t)out) | ((uintptr_t)skip)) & 15) == 0,
                    "alive_filter_block64: L must be a multiple of 4 and out / skip 16-byte aligned");
    ALIVE_CHECK_ARG(film_ld > 0 && t0 >= 0 && f0 >= 0, "alive_filter_block64: bad frame range");
    ALIVE_CHECK_ARG((double)BL * film_ld / L + 3.0 <= NFP, "alive_filter_block64: tile spans more than %d frames (L %d, frames %d)", NFP, L, film_ld);
    {
        static LdsOptIn optin;
        hipError_t e = H16 ? optin.ensure({(const void*)filter_block64_kernel<false, 4, H16>, (const void*)filter_block64_kernel<true, 4, H16>,
                                           (const void*)filter_block64_kernel<true, 2, H16>, (const void*)filter_block64_kernel<false, 4, true, true>}, LDS_BYTES)
                           : optin.ensure({(const void*)filter_block64_kernel<false, 4, H16>, (const void*)filter_block64_kernel<true, 4, H16>,
                                           (const void*)filter_block64_kernel<true, 2, H16>}, LDS_BYTES);
        if (e != hipSuccess) {
            alive_set_error("alive_filter_block64: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(e));
            return ALIVE_ERR_LAUNCH;
        }
    }
    const float ratio = (float)film_ld / (float)L;       // == window frames / window samples at this rate
    const int tiles = cdiv(L, TT);
    static const int nt_env = getenv("ALIVE_FB64_NT") ? atoi(getenv("ALIVE_FB64_NT")) : 0;
    if (nt_env ? nt_env == 2 : (int64_t)tiles * N <= 16) {
        // a handful of batch tiles: 128-column tiles (72 outputs) through the FIRST form, three times the blocks
        constexpr int TT2 = 128 - HALO, LDS2 = GUARD + 2 * (2 * 128 * ROWB) + NCONV * C * NFS * 8 + 128 * 8;
        filter_block64_kernel<true, 2, H16><<<dim3(cdiv(L, TT2), N), 256, LDS2, (hipStream_t)stream>>>(
            U, L, (const unsigned short*)W16, biases, film, film_rows, Lf, film_off, ratio, t0, f0, film_ld, skip, out, g_stamps64);
        ALIVE_CHECK_LAUNCH("alive_filter_block64");
        return ALIVE_OK;
    }
    // The first tile of every window reflects at t = 0 (per-lane fragment addresses): an instantiation and a launch of its own.  Its
    // address form is valid for every tile (no reflection happens further right, rows left of the image fall into the guard), only
    // slower -- so a problem that does not fill the chip anyway (the streaming ring: 4 tiles) runs ALL its tiles in that one launch
    // instead of two dependent ones.
    const bool small = (int64_t)tiles * N <= 256;
    filter_block64_kernel<true, 4, H16><<<dim3(small ? tiles : 1, N), 256, LDS_BYTES, (hipStream_t)stream>>>(
        U, L, (const unsigned short*)W16, biases, film, film_rows, Lf, film_off, ratio, t0, f0, film_ld, skip, out, g_stamps64);
    if constexpr (H16) {
        // the sweep form (see the kernel): columns [TT, L) in segments walked left to right; ALIVE_FB64_SWEEP=0 keeps the tiled form (A/B)
        static const bool sweep = !(getenv("ALIVE_FB64_SWEEP") && atoi(getenv("ALIVE_FB64_SWEEP")) == 0);
        if (sweep && tiles > 1 && !small) {
            const int rem = L - TT, t256 = cdiv(rem, BL);
            // segments per window: the fewest chip rounds x (tiles per segment + the warm-up tile)
            int best_s = 1;
            int64_t best_cost = -1;
            for (int sg = 1; sg <= (t256 < 64 ? t256 : 64); ++sg) {
                const int64_t cost = (int64_t)cdiv((int64_t)N * sg, 256) * (cdiv(t256, sg) + 1);
                if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_s = sg; }
            }
            const int seg_cols = cdiv(t256, best_s) * BL;
            filter_block64_kernel<false, 4, true, true><<<dim3(cdiv(rem, seg_cols), N), 256, LDS_BYTES, (hipStream_t)stream>>>(
                U, L, (const unsigned short*)W16, biases, film, film_rows, Lf, film_off, ratio, t0, f0, film_ld, skip, out, g_stamps64, TT, seg_cols);
            ALIVE_CHECK_LAUNCH("alive_filter_block64");
            return ALIVE_OK;
        }
    }
    if (tiles > 1 && !small)
        filter_block64_kernel<false, 4, H16><<<dim3(tiles - 1, N), 256, LDS_BYTES, (hipStream_t)stream>>>(
            U, L, (const unsigned short*)W16, biases, film, film_rows, Lf, film_off, ratio, t0, f0, film_ld, skip, out, g_stamps64);
    ALIVE_CHECK_LAUNCH("alive_filter_block64");
    return ALIVE_OK;
}
}  // namespace

extern "C" int alive_filter_block64_range(const float* U, int N, int L, const void* W16, const float* biases, const float* film,
                                          int film_rows, int Lf, int film_off, int t0, int f0, int film_ld, const float* skip,
                                          float* out, void* stream) {
    return filter_block64_impl<false>(U, N, L, W16, biases, film, film_rows, Lf, film_off, t0, f0, film_ld, skip, out, stream);
}
extern "C" int alive_filter_block64_range_fp16(const float* U, int N, int L, const void* W16, const float* biases, const float* film,
                                               int film_rows, int Lf, int film_off, int t0, int f0, int film_ld, const float* skip,
                                               float* out, void* stream) {
    return filter_block64_impl<true>(U, N, L, W16, biases, film, film_rows, Lf, film_off, t0, f0, film_ld, skip, out, stream);
}

ALIVE_F16_SAT_GETTER(alive_f16_sat_filter_mid)

