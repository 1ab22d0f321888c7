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
#include "planes_layout.h"
#include <stdlib.h>

#ifndef ALIVE_CONV_ABL
#define ALIVE_CONV_ABL 0             // ablation bits (timing only, WRONG results): 1 no epilogue, 2 B fragments read once per channel block, 4 no X DMA after the first,
                                     // 8 weight fragments loaded once, 16 no gelu, 32 FiLM without the interpolation, 64 no transposed plane leave
#endif

namespace {

constexpr int BN = 128;
constexpr int BKC = 32;           // channels per ci-block
constexpr int XROWS = 144;        // BN + max halo (4 taps x dilation 4)
constexpr int PITCH = 80;         // bytes per LDS row (32 bf16 + 16 B pad): 16-B aligned rows, conflict-free ds_read_b128
constexpr int XPLANE = XROWS * PITCH;

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p) { return *(const bf16x8*)p; }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int PROW = 64;          // PLANES = true: bytes per LDS row (32 channels bf16, no pad: 16-B chunks swizzled by the row)

// NP = number of bf16 planes per operand: 2 -> 3 MFMAs per product (~2^-16), 3 -> 6 MFMAs per product (~2^-24, fp32-grade)
// BM = 256: all 256 output channels of the filter's coarsest scale in ONE block -- a wave owns 64 rows x 128 columns (2 x 4 MFMA
// tiles, 128 accumulator registers, one block per CU).  Against two 128-row blocks per column tile that stages the X tile
// (global loads, split, LDS writes) once instead of twice, reads every activation fragment for two row tiles instead of one
// (0.33 instead of 0.67 LDS fragment reads per MFMA) and halves the X traffic from L2 / HBM.
// PLANES = true (BM = 128, NP = 2; round 3): the activation operand arrives ALREADY split, plane-packed with rows = time
// (AliveConv.Xp: the k-blocked planes of gemm_planes.hip / planes_layout.h, rows = n * Tin + t), so a 32-channel block of the X
// tile is ONE run of 144 consecutive 64-B row segments per plane that LDS-DMA copies straight into LDS -- no fp32 loads, no split, no ds_write in the
// loop.  By ablation (tools/experiments/README.md) the fp32 staging cost 0.27 - 0.35 ms of the 1.4 - 1.8 ms of a 256-channel
// k5 conv at 128 windows (its LDS writes alone 0.12 ms); the fragment READS per tap, 80 % of the LDS read traffic, cost nothing.
// The LDS image of a DMA is lane-linear, so the bank swizzle sits on the SOURCE address: 16-B chunk c of row r is stored at
// chunk position c ^ ((r >> 2) & 3), which keeps every ds_read_b128 of a fragment conflict-free at any tap shift.
// W22 (BM = 128, round 5): the four waves as 2 x 2 of 64 rows x 64 columns instead of four stacked 32 x 128 -- a B fragment from LDS then
// feeds two row tiles (0.33 instead of 0.67 fragment reads per MFMA: stacked, all four waves read the SAME 128 columns), an A fragment
// from L2 two column tiles instead of four (two waves ask for the same weight rows: the second request hits the CU's L1).
// NP = 1 (round 5, AliveConv.precision 3): plain fp16 operands, ONE MFMA per product (v_mfma_f32_32x32x16_f16) -- the weights are one
// fp16 plane (the third slab of module/_pack.py::pack_conv_split_h), the activations (Xp / Zp) one fp16 plane.  Used by the decoder for the six k = 5 convs of its 256-channel FilterBlock, whose inputs are
// gelu + FiLM outputs consumed by nothing else (1.2e-5 of waveform RMS error on the 450-frame fixture against the 1e-3 bar).  At one product per
// fragment pair the stacked 128-row form would read 1 KB of LDS per MFMA (the LDS peak): NP = 1 runs as the 2 x 2 form of BM = 128 (a B
// fragment feeds two row tiles; 144 registers, three blocks per CU).
template <int BM, int NP, bool PLANES = false, bool W22 = false>
__global__ __launch_bounds__(256, BM == 256 ? 1 : 2) void conv_split_kernel(AliveConv p, float film_ratio) {
    static_assert(!W22 || BM == 128, "the 2 x 2 wave arrangement is the 128-row tile's");
    // wave tile: (32 | 64) rows x (128 | 64) columns.  BM = 256 / 128: four waves stacked along the rows; BM = 64: 2 x 2.
    constexpr int NR = W22 ? 2 : (BM >= 128 ? 4 : 2);         // 32-column MFMA tiles per wave
    constexpr int MR = W22 ? 2 : (BM == 256 ? 2 : 1);         // 32-row MFMA tiles per wave

    // X tile, double buffered: [2 buffers][2 planes][XROWS][PITCH]
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (NP > 1 ? NP : 2) * XPLANE];      // (NP = 1: the epilogue's staging needs the 46 KB)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wrow = (BM >= 128 && !W22) ? wid : (wid >> 1);          // row block of this wave (32 * MR rows)
    const int wcol = (BM >= 128 && !W22) ? 0 : (wid & 1) * 64;        // first column of this wave
    const int lr = lane & 31, lh = lane >> 5;
    const int n = blockIdx.z, m0 = blockIdx.y * BM, t0 = blockIdx.x * BN;
    const int co_pad = (p.Co + 15) & ~15;
    const int K2 = p.KW * p.Ci_pad;
    const int ncb = p.Ci_pad / BKC;
    const int xrows = BN + (p.KW - 1) * p.dil;
    const unsigned short* W16 = (const unsigned short*)p.W;
    const float* Xn = p.X + (size_t)n * p.Ci * p.Tin;

    // ---- A fragments come STRAIGHT from global memory (the weights are L2-resident and each wave owns its 32 rows):
    // lane (row lr, half lh) loads the 16 B of k-step s, plane pl it feeds to the MFMA.  No LDS staging of the weights,
    // hence no block barrier per k-step: the only barrier left is the one that publishes the next X tile, once per
    // 32-channel block (KW taps x 24 MFMAs per wave).  One k-step is prefetched in registers.
    const unsigned short* Wrow[MR][NP];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        int grow = m0 + wrow * 32 * MR + mr * 32 + lr;
        grow = grow < co_pad ? grow : co_pad - 1;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) Wrow[mr][pl] = W16 + planes_at(pl, grow, 0, co_pad, K2) + lh * 8;      // k-blocked (planes_layout.h)
    }
    bf16x8 a_cur[MR][2][NP], a_nxt[MR][2][NP];    // [row tile][k16 step][plane]
    auto load_A = [&](int cb, int j, bf16x8 (&a)[MR][2][NP]) {
#ifdef ALIVE_CONV_ABL_SAMEA          // ablation (timing only, WRONG results): every k-step reads the weights of k-step 0 -- the L1 serves them
        const int kcol = 0 * (j + cb);
#else
        const int kcol = j * p.Ci_pad + cb * BKC;
#endif
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) a[mr][s2][pl] = *(const bf16x8*)(Wrow[mr][pl] + (size_t)kcol * co_pad + s2 * 16);   // kcol / 32 blocks of co_pad * 32
    };

    // ---- X staging: 3 (row, 8-channel group) items per thread per 32-channel block (144 rows x 4 groups = 576 items).  Everything that
    // does not depend on the channel block is hoisted: the clamped row offset, the validity mask, the LDS address.  Loads are
    // unconditional on clamped addresses (no branch per element; lanes = consecutive rows, so every load is coalesced along time), the
    // split uses the hardware f32->bf16 pack (v_cvt_pk_bf16_f32), and an item leaves as ONE 16-byte LDS write per plane.  Until round
    // 5 an item was a channel PAIR (4-byte writes at the row pitch of 80 B = 20 banks: consecutive rows cycle through 8 of the 32 banks,
    // a 4-way conflict on every write -- 43 % of this kernel's LDS cycles in profiles/r05_nets_pmc.json); a 16-byte write of
    // consecutive rows starts on those 8 banks and covers 4 each: all 32, conflict-free.  Same values, same LDS image.
    constexpr int XIT = 3;
    float x_reg[XIT][8];
    int x_off[XIT];               // group*8*Tin + clamped tin  (element offset inside the channel block)
    unsigned x_ok = 0;            // bit it: row inside the tile and inside the signal
    unsigned x_in = 0;            // bit it: the item exists (it * 256 + tid < 576)
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        int i = it * 256 + tid;
        const bool in = i < XROWS * 4;
        i = in ? i : 0;
        int r = i % XROWS, grp = i / XROWS;
        int tin = t0 - p.pad_left + r;
        if (tin < 0 && p.pad_mode != 0) tin = -tin;
        bool ok = r < xrows && tin >= 0 && tin < p.Tin;
        tin = tin < 0 ? 0 : (tin < p.Tin ? tin : p.Tin - 1);
        x_off[it] = grp * 8 * p.Tin + tin;
        x_ok |= ((ok && in) ? 1u : 0u) << it;
        x_in |= (in ? 1u : 0u) << it;
    }
    const bool ragged_ci = (p.Ci % BKC) != 0;
    auto load_X = [&](int cb) {
        const float* Xc = Xn + (size_t)cb * BKC * p.Tin;
        const bool tail = ragged_ci && (cb + 1) * BKC > p.Ci;          // block-uniform
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            if (!((x_in >> it) & 1u)) continue;
            if (!tail) {
#pragma unroll
                for (int c = 0; c < 8; ++c) x_reg[it][c] = Xc[x_off[it] + c * p.Tin];
            } else {
                const int grp = (it * 256 + tid) / XROWS;
                const int ci = cb * BKC + grp * 8;
#pragma unroll
                for (int c = 0; c < 8; ++c) x_reg[it][c] = ci + c < p.Ci ? Xc[x_off[it] + c * p.Tin] : 0.0f;
            }
        }
    };
    auto store_X = [&](int buf) {
        unsigned char* Xs = smem + buf * NP * XPLANE;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            if (!((x_in >> it) & 1u)) continue;
            const int i = it * 256 + tid;
            const int r = i % XROWS, grp = i / XROWS;
            const bool ok = (x_ok >> it) & 1u;
            u32x4 hv, lv, tv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = ok ? x_reg[it][2 * e] : 0.0f, x1 = ok ? x_reg[it][2 * e + 1] : 0.0f;
                bf16x2_t hp = {(__bf16)x0, (__bf16)x1};
                const unsigned h = NP == 1 ? pack_f16x2(x0, x1) : __builtin_bit_cast(unsigned, hp);          // (NP = 1: the one plane is fp16)
                const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
                bf16x2_t lp = {(__bf16)r0, (__bf16)r1};
                const unsigned l = __builtin_bit_cast(unsigned, lp);
                hv[e] = h;
                lv[e] = l;
                if (NP == 3) {
                    const float q0 = r0 - __uint_as_float(l << 16), q1 = r1 - __uint_as_float(l & 0xffff0000u);
                    bf16x2_t tp = {(__bf16)q0, (__bf16)q1};
                    tv[e] = __builtin_bit_cast(unsigned, tp);
                }
            }
            unsigned char* dst = Xs + r * PITCH + grp * 16;
            *(u32x4*)dst = hv;
            if (NP >= 2) *(u32x4*)(dst + XPLANE) = lv;
            if (NP == 3) *(u32x4*)(dst + 2 * XPLANE) = tv;
        }
    };

    // ---- PLANES: DMA pieces of this wave (piece q = wid + 4 i < 9 NP: plane q / 9, 16-row group q % 9) ----
    unsigned dma_src[5];          // byte offset of this lane's 16 B inside Xp for channel block 0
    [[maybe_unused]] const size_t cols_pad = PLANES ? (((size_t)p.N * p.Tin + 127) & ~(size_t)127) : 0;
    if constexpr (PLANES) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = wid + 4 * i;
            const int pl = q / 9, r = (q % 9) * 16 + (lane >> 2);
            int tin = t0 - p.pad_left + r;
            tin = tin < 0 ? -tin : tin;                                  // reflect-left (pad_mode 1; checked on the host)
            tin = tin < p.Tin ? tin : p.Tin - 1;                         // rows past the signal only feed columns >= Tout
            const int chunk = (lane & 3) ^ ((r >> 2) & 3);
            dma_src[i] = (unsigned)((planes_at(pl, (int64_t)n * p.Tin + tin, 0, cols_pad, p.Ci_pad) + chunk * 8) * 2);
        }
    }
    auto dma_X = [&](int cb, int buf) {
        const unsigned char* base = (const unsigned char*)p.Xp + (size_t)cb * cols_pad * (BKC * 2);      // one k-block further
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = wid + 4 * i;
            if (q < 9 * NP)
                __builtin_amdgcn_global_load_lds((gptr_t)(base + dma_src[i]),
                                                 (lptr_t)(smem + (buf * NP + q / 9) * XPLANE + (q % 9) * 1024), 16, 0, 0);
        }
    };

    // ---- accumulators start at the bias ----
    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        f32x16 b16;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = m0 + wrow * 32 * MR + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            b16[r] = (p.bias != nullptr && row < p.Co) ? p.bias[row] : 0.0f;
        }
#pragma unroll
        for (int nn = 0; nn < NR; ++nn) acc[mr][nn] = b16;
    }

    // weights are prefetched TWO k-steps ahead (fragment-shaped L2 reads have a long tail under load)
    bf16x8 a_nx2[MR][2][NP];
    int pcb = 0, pj = 0;                                       // (block, tap) of the next fragment set to fetch
    auto advance = [&]() { if (++pj == p.KW) { pj = 0; ++pcb; } };
    if constexpr (PLANES) dma_X(0, 0); else load_X(0);
    load_A(0, 0, a_cur);
    advance();
    if (pcb < ncb) load_A(pcb, pj, a_nxt);
    advance();
    if constexpr (!PLANES) store_X(0);
    __syncthreads();
#ifdef ALIVE_CONV_PRIO
    __builtin_amdgcn_s_setprio(ALIVE_CONV_PRIO);              // the k-loop of this wave before the epilogue of a co-resident block's
#endif
    for (int cb = 0; cb < ncb; ++cb) {
        const bool more_cb = cb + 1 < ncb;
        if (ALIVE_CONV_ABL & 8) pcb = ncb;                     // no further weight loads
        if constexpr (PLANES) {
            if (more_cb && !(ALIVE_CONV_ABL & 4)) dma_X(cb + 1, (cb + 1) & 1);          // lands under this block's taps; the barrier below waits for it
        } else {
            if (more_cb) load_X(cb + 1);                       // in flight under this block's taps
        }
        const unsigned char* Xs = smem + (cb & 1) * NP * XPLANE;
#if ALIVE_CONV_ABL & 2
        bf16x8 bf[NP][NR];
#endif
        for (int j = 0; j < p.KW; ++j) {
            if (pcb < ncb) load_A(pcb, pj, a_nx2);
            advance();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#if !(ALIVE_CONV_ABL & 2)
                bf16x8 bf[NP][NR];
#endif
                if ((ALIVE_CONV_ABL & 2) && (j > 0 || s2 > 0)) {
                } else
                if constexpr (PLANES) {
                    const int rho = wcol + lr + j * p.dil;             // + nn * 32 does not change (row >> 2) & 3
                    const unsigned char* b = Xs + rho * PROW + (((2 * s2 + lh) ^ ((rho >> 2) & 3)) << 4);
#pragma unroll
                    for (int nn = 0; nn < NR; ++nn)
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) bf[pl][nn] = lds_frag(b + nn * 32 * PROW + pl * XPLANE);
                } else {
#pragma unroll
                for (int nn = 0; nn < NR; ++nn) {
                    const unsigned char* b = Xs + (wcol + nn * 32 + lr + j * p.dil) * PITCH + s2 * 32 + lh * 16;
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) bf[pl][nn] = lds_frag(b + pl * XPLANE);
                }
                }
                // all plane products (i, j) with i + j <= NP - 1, smallest terms first
#pragma unroll
                for (int nn = 0; nn < NR; ++nn) {
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                        for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
                            for (int i = 0; i <= sum; ++i)
                                acc[mr][nn] = NP == 1 ? mfma_f16(a_cur[mr][s2][i], bf[sum - i][nn], acc[mr][nn])
                                                      : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[mr][s2][i], bf[sum - i][nn], acc[mr][nn], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) { a_cur[mr][s2][pl] = a_nxt[mr][s2][pl]; a_nxt[mr][s2][pl] = a_nx2[mr][s2][pl]; }
        }
        if constexpr (!PLANES) { if (more_cb) store_X((cb + 1) & 1); }
        __syncthreads();          // next X tile visible (vmcnt(0) covers the DMA); this one is free to be overwritten one block later
    }

    if (ALIVE_CONV_ABL & 1) {
        float sx = 0.0f;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nn = 0; nn < NR; ++nn) sx += acc[mr][nn][lane & 15];
        if (sx == 12345.678f && p.Y != nullptr) p.Y[0] = sx;
        return;
    }
#ifdef ALIVE_CONV_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // ---- epilogue: accumulators -> LDS -> cooperative row-wise pass ----
    // Staging the tile through LDS turns the MFMA layout (a lane owns 16 scattered rows of one column) into
    // whole rows: every global access of the epilogue (residual, skip, Y, Z) is a 16-B vector per thread on
    // 512-B contiguous row segments, and the element code exists once in a rolled loop instead of 64 times.
    constexpr int PR = 64;                           // rows per pass
    constexpr int NPASS = BM / 64;                   // passes
    constexpr int CP = BN + 4;                       // fp32 pitch of the staged tile
    float* Ct = (float*)smem;                        // [PR][CP]
    float* Ft = Ct + PR * CP;                        // [PR][2][FILM_NF]
    static_assert((PR * CP + PR * 2 * FILM_NF) * 4 <= (int)sizeof(smem), "epilogue staging fits");
    int f_lo = 0, nf = 0;
    if (p.Z != nullptr || p.Zp != nullptr) film_tile_range(p, film_ratio, t0, BN, f_lo, nf);       // fit is checked on the host
    const bool vec = (p.up == 1) && ((p.Tout & 3) == 0);
    // Plane-packed second output (AliveConv.Zp): the modulated tile goes back into the staged tile and leaves it TRANSPOSED, 8
    // channels x 1 column per thread.  Column-wise reads of [row][column] floats put rows 16 apart on the same banks, so the
    // staged tile is column-swizzled in units of 8 columns by (row >> 4) & 3 (a multiple of the 4-column vectors of the
    // row-wise pass): zsw_on = 8 when Zp is set, 0 otherwise (the fp32 outputs keep their layout).
    const int zsw_on = p.Zp != nullptr ? 3 : 0;
    auto zsw = [&](int row_local) { return ((row_local >> 4) & zsw_on) << 3; };
    const bool upvec = p.up >= 4 && (p.up & (p.up - 1)) == 0 && p.up <= PR && (p.Co & (p.up - 1)) == 0;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        if ((MR == 2 ? wrow : (wrow >> 1)) == ps) {  // wave-uniform: the wave(s) that own the 64 rows of this pass deposit their tiles
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nn = 0; nn < NR; ++nn)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                    {
                        const int rl = (MR == 2 ? mr : (wrow & 1)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        Ct[rl * CP + ((wcol + nn * 32 + lr) ^ zsw(rl))] = acc[mr][nn][r];
                    }
        }
        if (p.Z != nullptr || p.Zp != nullptr) {
            for (int e = tid; e < PR * 2 * FILM_NF; e += 256) {
                int f = e % FILM_NF, sel = (e / FILM_NF) & 1, pr = e / (2 * FILM_NF);
                int row = m0 + ps * 64 + pr;
                float v = 0.0f;
                if (row < p.Co && f < nf) {
                    const int ld = p.film_ld ? p.film_ld : p.Lf;
                    int c = f_lo + f - p.film_f0;                    // frame f_lo + f of the window -> column of the film tensor
                    c = c < 0 ? 0 : (c < ld ? c : ld - 1);
                    v = p.film[((size_t)n * p.film_rows + (sel == 0 ? p.film_scale_row : p.film_shift_row) + row) * ld + c];
                }
                Ft[e] = v;
            }
        }
        __syncthreads();
        if (vec) {
            // PR*BN/4/256 = 8 iterations, fully unrolled: the residual / skip loads of all eight 16-B groups are in flight
            // together (rolled, each iteration exposed a full L2 round trip and the epilogue cost as much as the GEMM)
#pragma unroll
            for (int g = tid; g < PR * (BN / 4); g += 256) {
                const int pr = g >> 5, c4 = (g & 31) * 4;
                const int row = m0 + ps * 64 + pr;
                const int t = t0 + c4;
                if (row >= p.Co || t >= p.Tout) continue;
                f32x4 v = *(const f32x4*)&Ct[pr * CP + (c4 ^ zsw(pr))];
                if (p.act == 1) {                                  // two values per instruction on the packed fp32 pipe
                    const f32x2 g0 = gelu_fast2(f32x2{v[0], v[1]}), g1 = gelu_fast2(f32x2{v[2], v[3]});
                    v = f32x4{g0[0], g0[1], g1[0], g1[1]};
                }
                else if (p.act == 2) { for (int q = 0; q < 4; ++q) v[q] = expf(v[q]); }
                if (p.post_add != nullptr) v = v + p.post_add[row];
                if (p.ch_scale != nullptr) v = v * p.ch_scale[row];
                const size_t o = ((size_t)n * p.Co + row) * p.Tout + t;
                if (p.residual != nullptr) v = v + *(const f32x4*)(p.residual + o);
                if (p.skip != nullptr) v = v + *(const f32x4*)(p.skip + o);
                if (p.Y != nullptr) *(f32x4*)(p.Y + o) = v;
                if (p.Z != nullptr || p.Zp != nullptr) {
                    const float* fs = Ft + pr * 2 * FILM_NF - f_lo;
                    f32x4 z;
#if ALIVE_CONV_ABL & 16
                    const float gv[4] = {v[0], v[1], v[2], v[3]};
#else
                    const f32x2 g0 = gelu_fast2(f32x2{v[0], v[1]}), g1 = gelu_fast2(f32x2{v[2], v[3]});
                    const float gv[4] = {g0[0], g0[1], g1[0], g1[1]};
#endif
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#if ALIVE_CONV_ABL & 32
                        z[q] = gv[q] * fs[q] + fs[FILM_NF + q];
#else
                        Lerp lp = lerp_coord(t + q + p.film_t0, film_ratio, p.Lf);
                        float sc = lerp_apply(lp, fs[lp.i0], fs[lp.i1]);
                        float sh = lerp_apply(lp, fs[FILM_NF + lp.i0], fs[FILM_NF + lp.i1]);
                        z[q] = gv[q] * sc + sh;
#endif
                    }
                    if (p.Z != nullptr) *(f32x4*)(p.Z + o) = z;
                    if (p.Zp != nullptr) *(f32x4*)&Ct[pr * CP + (c4 ^ zsw(pr))] = z;      // in place: this thread's own four values
                }
            }
            if (p.Zp != nullptr && !(ALIVE_CONV_ABL & 64)) {
                // transposed leave: thread = (column, 8 channels); 8 lanes write the 128 B of a column's 64 channels per plane
                __syncthreads();
                unsigned short* Zp = (unsigned short*)p.Zp;
                const int co_pad32 = (p.Co + 31) & ~31;
                const size_t zcols_pad = ((size_t)p.N * p.Tout + 127) & ~(size_t)127;
#pragma unroll
                for (int e = tid; e < BN * 8; e += 256) {
                    const int chunk = e & 7, t_l = e >> 3;
                    const int t = t0 + t_l, row0 = m0 + ps * 64 + chunk * 8;
                    if (t >= p.Tout || row0 >= p.Co) continue;
                    const float* cp = Ct + (chunk * 8) * CP + (t_l ^ (((chunk >> 1) & 3) << 3));
                    unsigned hi[4], lo[4];
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) {
                        const float q0 = cp[(2 * k2) * CP], q1 = cp[(2 * k2 + 1) * CP];
                        bf16x2_t hp = {(__bf16)q0, (__bf16)q1};
                        hi[k2] = NP == 1 ? pack_f16x2(q0, q1) : __builtin_bit_cast(unsigned, hp);
                        bf16x2_t lp2 = {(__bf16)(q0 - __uint_as_float(hi[k2] << 16)), (__bf16)(q1 - __uint_as_float(hi[k2] & 0xffff0000u))};
                        lo[k2] = __builtin_bit_cast(unsigned, lp2);
                    }
                    unsigned short* dst = Zp + planes_at(0, (int64_t)n * p.Tout + t, row0, zcols_pad, co_pad32);
                    *(u32x4*)dst = u32x4{hi[0], hi[1], hi[2], hi[3]};
                    if (NP >= 2) *(u32x4*)(dst + zcols_pad * co_pad32) = u32x4{lo[0], lo[1], lo[2], lo[3]};     // (NP = 1: the consumer reads one plane)
                }
            }
        } else if (upvec) {
            // ConvTranspose1d(k == stride == up), up a multiple of 4 dividing PR: the rows (co, j) of one output channel hold
            // the up * BN CONSECUTIVE samples Y[co][t0 * up ...] in [t][j] order -- gather them from the staged tile and store
            // 16-B vectors (the scalar path below scatters 4-B stores at a stride of up floats)
            const int sh = __builtin_ctz((unsigned)p.up);
            const size_t Lout = (size_t)p.Tout * p.up;
#pragma unroll
            for (int g = tid; g < PR * (BN / 4); g += 256) {
                const int e = g * 4;                                   // element of the pass: [co_l][t_l][j]
                const int co_l = e >> (7 + sh), rem = e & ((BN << sh) - 1);
                const int t_l = rem >> sh, j0 = rem & (p.up - 1);
                const int pr = (co_l << sh) + j0;
                const int row = m0 + ps * 64 + pr;
                const int t = t0 + t_l;
                if (row >= p.Co || t >= p.Tout) continue;
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = Ct[(pr + q) * CP + t_l];
                    if (p.act == 1) x = gelu_fast(x);
                    else if (p.act == 2) x = expf(x);
                    if (p.post_add != nullptr) x += p.post_add[row + q];
                    if (p.ch_scale != nullptr) x *= p.ch_scale[row + q];
                    v[q] = x;
                }
                *(f32x4*)(p.Y + ((size_t)n * (p.Co >> sh) + (row >> sh)) * Lout + ((size_t)t << sh) + j0) = v;
            }
        } else if (p.up > 1) {
            // any other ConvTranspose1d rate (10 on the coarsest filter scale): walk the OUTPUT order [co][t][j] of the
            // channels this pass touches, so that consecutive lanes store consecutive samples; rows of the first / last
            // channel that belong to the neighbouring pass are skipped
            const unsigned inv_up = (unsigned)((0x100000000ull + (unsigned)p.up - 1) / (unsigned)p.up);   // exact floor(x / up), x < 2^25
            const int R0 = m0 + ps * 64;
            const int c_first = (int)__umulhi((unsigned)R0, inv_up);
            int R1 = R0 + PR - 1;
            R1 = R1 < p.Co ? R1 : p.Co - 1;
            const int nch = (int)__umulhi((unsigned)R1, inv_up) - c_first + 1;
            const int per_ch = BN * p.up;
            const size_t Lout = (size_t)p.Tout * p.up;
            const int Cq = (int)__umulhi((unsigned)p.Co, inv_up);
#pragma unroll 1
            for (int ch = 0; ch < nch; ++ch) {
                const int co = c_first + ch;
                float* yrow = p.Y + ((size_t)n * Cq + co) * Lout + (size_t)t0 * p.up;
#pragma unroll 2
                for (int e = tid; e < per_ch; e += 256) {
                    const int t_l = (int)__umulhi((unsigned)e, inv_up), jj = e - t_l * p.up;
                    const int row = co * p.up + jj;
                    if (row < R0 || row > R1 || t0 + t_l >= p.Tout) continue;
                    float v = Ct[(row - R0) * CP + t_l];
                    if (p.act == 1) v = gelu_fast(v);
                    else if (p.act == 2) v = expf(v);
                    if (p.post_add != nullptr) v = v + p.post_add[row];
                    if (p.ch_scale != nullptr) v = v * p.ch_scale[row];
                    yrow[e] = v;
                }
            }
        } else {
#pragma unroll 1
            for (int g = tid; g < PR * BN; g += 256) {
                const int pr = g >> 7, c = g & 127;
                const int row = m0 + ps * 64 + pr;
                const int t = t0 + c;
                if (row >= p.Co || t >= p.Tout) continue;
                float v = Ct[pr * CP + c];
                if (p.act == 1) v = gelu_fast(v);
                else if (p.act == 2) v = expf(v);
                if (p.post_add != nullptr) v = v + p.post_add[row];
                if (p.ch_scale != nullptr) v = v * p.ch_scale[row];
                const size_t o = ((size_t)n * p.Co + row) * p.Tout + t;
                if (p.residual != nullptr) v = v + p.residual[o];
                if (p.skip != nullptr) v = v + p.skip[o];
                if (p.Y != nullptr) p.Y[o] = v;
                if (p.Z != nullptr) {
                    const float* fs = Ft + pr * 2 * FILM_NF - f_lo;
                    Lerp lp = lerp_coord(t + p.film_t0, film_ratio, p.Lf);
                    float sc = lerp_apply(lp, fs[lp.i0], fs[lp.i1]);
                    float sh = lerp_apply(lp, fs[FILM_NF + lp.i0], fs[FILM_NF + lp.i1]);
                    p.Z[o] = gelu_fast(v) * sc + sh;
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
        const double span = (double)(d->film_ld ? d->film_ld : d->Lf) / (double)d->Tout * BN + 3.0;
        ALIVE_CHECK_ARG(span <= FILM_NF, "alive_conv1d(split): FiLM second output needs Tout >= ~8 Lf (got Lf %d, Tout %d)", d->Lf, d->Tout);
    }
    ALIVE_CHECK_ARG(d->act >= 0 && d->act <= 2, "alive_conv1d(split): activation %d not available on the split kernel", d->act);
    if (d->Zp) {
        const double span = (double)(d->film_ld ? d->film_ld : d->Lf) / (double)d->Tout * BN + 3.0;
        ALIVE_CHECK_ARG(span <= FILM_NF && d->film, "alive_conv1d(split): plane second output needs FiLM rows and Tout >= ~8 Lf");
        ALIVE_CHECK_ARG(d->up == 1 && (d->Tout & 3) == 0 && d->Co % 64 == 0 && d->Co > 64 && (d->precision == 1 || d->precision == 3) &&
                        (((uintptr_t)d->Zp) & 15) == 0,
                        "alive_conv1d(split): plane second output needs up 1, Tout %% 4 == 0, Co a multiple of 64 above 64, precision 1 or 3");
    }
    if (d->Xp) {
        ALIVE_CHECK_ARG((d->precision == 1 || d->precision == 3) && d->Co > 64 && d->Ci % BKC == 0 && d->Ci_pad == d->Ci && (d->pad_mode == 1 || d->pad_left == 0) &&
                        (((uintptr_t)d->Xp) & 15) == 0 && d->Tout <= d->Tin &&
                        (int64_t)(d->Tout - 1) * d->stride + (int64_t)(d->KW - 1) * d->dil - d->pad_left < d->Tin,
                        "alive_conv1d(split): plane input needs precision 1 or 3, Co > 64, Ci a multiple of 32, reflect-left padding and no valid "
                        "output column that reads past the signal (the kernel clamps rows >= Tin instead of zero-filling them)");
    }
    ALIVE_CHECK_ARG(d->Tout <= d->Tin + d->pad_left, "alive_conv1d(split): Tout");
    // measured (tools/bench_conv256.py, 128 windows x 4500 columns): the 256-row tile is 12 - 37 % SLOWER than two 128-row blocks
    // (k5 + FiLM + residual 2.11 against 1.72 ms, 1x1 1.15 against 0.84): at one block per CU nothing covers the LDS / L2
    // latencies of the single wave per SIMD, and no other block's main loop runs under the four epilogue passes.  Off by
    // default; ALIVE_CONV_TILE256=1 selects it (same results bit for bit).
    static const bool tile256 = getenv("ALIVE_CONV_TILE256") != nullptr && atoi(getenv("ALIVE_CONV_TILE256")) != 0;
    static const bool w22 = getenv("ALIVE_CONV_W22") != nullptr && atoi(getenv("ALIVE_CONV_W22")) != 0;      // A/B: 2 x 2 waves (plane input)
    if (d->precision == 3) {
        // plain fp16 (one plane per operand): the 2 x 2 form of the 128-row tile.  Measured (tools/bench_conv256.py, 128 windows x 4500 columns,
        // k5 + FiLM + residual + Y / k5 + FiLM / 1x1 + FiLM + Y / k5 + residual + Y): 0.895 / 0.697 / 0.548 / 0.661 ms against the two-plane
        // form's 1.353 / 1.182 / 0.664 / 1.063; as one 256-row block per column tile (256 registers at two blocks per CU, 122 of them
        // spilled in the four epilogue passes) 1.625 / 1.319 / 1.239 / 0.747 -- not instantiated.
        if (d->Co > 64) {
            dim3 g(cdiv(d->Tout, BN), cdiv(d->Co, 128), d->N);
            if (d->Xp) conv_split_kernel<128, 1, true, true><<<g, 256, 0, s>>>(*d, ratio);
            else conv_split_kernel<128, 1, false, true><<<g, 256, 0, s>>>(*d, ratio);
        } else {
            dim3 g(cdiv(d->Tout, BN), 1, d->N);
            conv_split_kernel<64, 1><<<g, 256, 0, s>>>(*d, ratio);
        }
    } else if (d->Co > 128 && d->Co % 256 == 0 && d->precision == 1 && tile256 && !d->Xp) {
        dim3 g(cdiv(d->Tout, BN), d->Co / 256, d->N);
        conv_split_kernel<256, 2><<<g, 256, 0, s>>>(*d, ratio);
    } else if (d->Co > 64) {
        dim3 g(cdiv(d->Tout, BN), cdiv(d->Co, 128), d->N);
        if (d->precision == 2) conv_split_kernel<128, 3><<<g, 256, 0, s>>>(*d, ratio);
        else if (d->Xp && w22) conv_split_kernel<128, 2, true, true><<<g, 256, 0, s>>>(*d, ratio);
        else if (d->Xp) conv_split_kernel<128, 2, true><<<g, 256, 0, s>>>(*d, ratio);
        else conv_split_kernel<128, 2><<<g, 256, 0, s>>>(*d, ratio);
    } else {
        dim3 g(cdiv(d->Tout, BN), 1, d->N);
        if (d->precision == 2) conv_split_kernel<64, 3><<<g, 256, 0, s>>>(*d, ratio);
        else conv_split_kernel<64, 2><<<g, 256, 0, s>>>(*d, ratio);
    }
    ALIVE_CHECK_LAUNCH("alive_conv1d(split)");
    return ALIVE_OK;
}

ALIVE_F16_SAT_GETTER(alive_f16_sat_conv_split)
