// Frame-rate GEMMs of the ConvNeXt stacks on plane-packed operands (include/alive_vc.h "plane-packed activations").
//
//   D[co][col] = bias[co] + sum_k W[co][k] * X[col][k]          co < Co, col = n*T + t < N*T, k < Ci
//
// Both operands are stored as NP bf16 planes (v = p0 + p1 (+ p2)), K-BLOCKED (planes_layout.h): element (plane, row, k) of an
// operand with R padded rows sits at ((plane * K/32 + k/32) * R + row) * 32 + k % 32, so the 64-byte k-segments of consecutive
// rows are consecutive in memory and one K-step of a 128-row tile plane is ONE contiguous 8-KB run that LDS-DMA
// (`global_load_lds_dwordx4`) copies straight into LDS in 1-KB pieces of whole cache lines: no VGPR round trip, no conversion
// and no ds_write in the loop, which is what bound conv_split.hip on these shapes (one 32-channel block of a 1x1 conv is only
// 24 MFMAs per wave between two barriers, and the fp32 activations had to be loaded, split and stored behind them).
// (Until round 4 the planes were row-major with k contiguous: a piece was 16 row segments of 64 B, each HALF of a 128-B line
// whose other half the next K-step asked for again -- twice the L2 requests for the same bytes; these kernels ran 5 - 12 %
// slower on it, tools/experiments/README.md.)
// The product of two split values keeps the plane pairs (i, j) with i + j <= NP - 1:
//   NP = 2: 3 MFMAs ("bf16x3", ~2^-16 per product)    NP = 3: 6 MFMAs ("bf16x6", fp32-grade)
//
// Block = 4 waves (2 x 2), tile 128 (co) x 128 (col), wave tile 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_bf16 tiles.
// K advances 32 per step; a step's operands (2 x NP x 8 KB) live in one slot of an NS-slot LDS ring.  After the
// barrier of step s the slot of step s is free and the DMA of step s + NS goes into it, so NS - 2 steps are always in
// flight behind the one that must have landed (counted vmcnt, never 0 in the steady state).
// LDS image of a tile plane: [128 rows][64 B], filled in 1-KB pieces of 16 rows; the LDS side of an LDS-DMA is
// lane-linear, so the bank swizzle sits on the SOURCE address: 16-B chunk c of row r is stored at chunk
// c ^ ((r >> 2) & 3), which makes every ds_read_b128 of an MFMA fragment conflict-free (16-lane groups, 256-B bank rows).
#include "conv_epilogue.h"
#include "planes_layout.h"

namespace {

constexpr int GM = 128, GN = 128, GK = 32;
constexpr int PLANE_BYTES = GM * GK * 2;          // 8 KB: one operand plane of one step

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    // gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14 ; unused counters at their maximum
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void wait_lgkmcnt0() { __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14)); }

// Where K-step s of a tile's operands starts, relative to step 0 (elements).  W and the packed B planes are k-blocked: one block
// further is one padded operand further (a_ks, b_blk).  Custom row placement (AliveGemm.b_row != 0) walks `ncb` k-blocks per tap
// and then moves one row segment (32 elements) on: ncb = 1, b_blk = 0 for rows that are k-contiguous in memory (the STFT's signal).
struct GemmWalk {
    int64_t a_ks, b_blk, b_tap;
    int ncb;
    int rg, n_ct;        // tile order (tile_of below): row tiles per group, column tiles
};
// Tile order: ROW GROUPS of rg row tiles, inside a group column tile by column tile.  The blocks an XCD runs together (32 CUs,
// consecutive tiles of its contiguous chunk of this order) then share rg weight tiles -- chosen to stay resident in the XCD's 4-MB
// L2 -- and 32 / rg activation panels that stream through it.  With all row tiles inside one column tile (rounds 2 - 3) an XCD
// streamed the WHOLE weight matrix per column tile: 12.7 MB at 512 -> 4128 x 3 planes, nothing of it still in L2 when the next
// column tile asked again (tools/pmc_mem.sh: 28 % of the L2 requests missed).
__device__ __forceinline__ void tile_of(int v, int n_mt, const GemmWalk& g, int& mt, int& ct) {
    const int gs = g.rg * g.n_ct;
    const int grp = v / gs, rem = v - grp * gs;
    const int left = n_mt - grp * g.rg;
    const int rows = left < g.rg ? left : g.rg;
    ct = rem / rows;
    mt = grp * g.rg + (rem - ct * rows);
}
struct StepWalk {
    int64_t a = 0, b = 0;
    int blk = 0;
    __device__ __forceinline__ void reset() { a = 0; b = 0; blk = 0; }
    __device__ __forceinline__ void advance(const GemmWalk& g) {
        a += g.a_ks;
        b += g.b_blk;
        if (++blk == g.ncb) { blk = 0; b += g.b_tap - (int64_t)g.ncb * g.b_blk; }
    }
};

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}
// the plane format of an NP-plane operand: NP = 1 is ONE fp16 plane (plain operands, round 5), NP >= 2 split bf16 planes
template <int NP>
__device__ __forceinline__ unsigned pack_plane2(float a, float b, bool count = true) { return NP == 1 ? pack_f16x2(a, b, count) : pack_bf16x2(a, b); }
// one plane of a split: the packed pair, (a, b) left as the remainders.  F16S: fp16 planes (AliveGemm.f16s), else bf16 (NP = 1: plain fp16)
// count: fp16 saturations are counted for real outputs only (a column past the end of the problem is computed from whatever the padding of
// its operand buffer holds and is skipped on the way out)
template <int NP, bool F16S>
__device__ __forceinline__ unsigned split_step(float& a, float& b, bool count = true) {
    if constexpr (F16S) {
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
        const unsigned h = pack_f16x2(a, b, count);
        const f16x2_t hv = __builtin_bit_cast(f16x2_t, h);
        a -= (float)hv[0];
        b -= (float)hv[1];
        return h;
    } else {
        const unsigned h = pack_plane2<NP>(a, b, count);
        a -= __uint_as_float(h << 16);
        b -= __uint_as_float(h & 0xffff0000u);
        return h;
    }
}

// ---- epilogue, straight from the accumulators (shared by the one-tile and the persistent kernel) ----
// stage: wave-private LDS (NP x 8 KB: [plane][64 columns][64 channels] bf16) for the plane-packed output, or nullptr.  With it
// the planes leave as 16-byte pieces, 64 lanes covering 16 columns x the 64 B of a k-block (the tile is written to LDS in the
// accumulator layout -- 8 B per lane, 16-B chunk index XOR-ed with column & 7 -- and read back transposed); without it a lane
// stores its 8-byte pieces directly, one per column and store instruction: 64 separate 64-B lines touched per instruction.
// By ablation (tools/experiments/README.md) the epilogue was 41 % of a 512 -> 1536 + GELU -> planes layer.
template <int NP, int ACT, bool F16S = false>
__device__ __forceinline__ void gemm_epilogue(const AliveGemm& p, f32x16 (&acc)[2][2], int m0, int64_t c0, int wr, int wc, int lr,
                                              int lh, int64_t cols, int64_t cols_pad, int co_pad32, unsigned char* stage = nullptr,
                                              unsigned char* stage_small = nullptr) {
    // A lane holds, per 32 x 32 tile, one column and 16 rows in 4 groups of 4 consecutive rows (8 g + 4 lh + e).
    // Offsets are 32-bit (checked on the host) against uniform bases; per-row vectors (bias, post_add, ch_scale) come
    // as clamped loads, and all residual values of a 32-row half are requested before its first store (Y may alias the
    // residual, so the compiler may not hoist them).  Tiles that lie completely inside Co take the store path without
    // per-row tests.
    if constexpr (ACT == 3) {
        // argmax over the rows: the wave's 64 x 64 sub-tile leaves one (value, row) candidate per column.  A lane walks
        // its 32 rows in ascending order, the two half-waves of a column (rows 8 g + 4 lh + e) meet through one
        // cross-lane exchange; value = accumulator + bias, bitwise what the Y path would have stored.
        float bv[2] = {-INFINITY, -INFINITY};
        int bi[2] = {0x7fffffff, 0x7fffffff};
        auto take = [](float v, int vi, float b, int i) { return (v > b) || (v == b && vi < i) || (v != v && b == b); };
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int rbase = m0 + wr * 64 + ti * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                const float b = p.bias != nullptr ? p.bias[row < p.Co ? row : p.Co - 1] : 0.0f;
                const float osc = F16S ? p.wscale[0] * p.in_unscale : 1.0f;
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    const float x = F16S ? fmaf(acc[ti][tj][r], osc, b) : acc[ti][tj][r] + b;
                    if (row < p.Co && take(x, row, bv[tj], bi[tj])) { bv[tj] = x; bi[tj] = row; }
                }
            }
        }
        const int blk = (m0 + wr * 64) >> 6;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const float ov = __shfl_xor(bv[tj], 32);
            const int oi = __shfl_xor(bi[tj], 32);
            if (take(ov, oi, bv[tj], bi[tj])) { bv[tj] = ov; bi[tj] = oi; }
            const int64_t col = c0 + wc * 64 + tj * 32 + lr;
            if (lh == 0 && col < cols && m0 + wr * 64 < p.Co) {
                p.arg_val[(size_t)blk * cols + col] = bv[tj];
                p.arg_idx[(size_t)blk * cols + col] = bi[tj];
            }
        }
        return;
    }
    if constexpr (ACT == 4) {
        // Magnitudes of (re, im) row pairs -> plane-packed channels (include/alive_vc.h, AliveGemm.act == 4).  The wave's 64 rows are
        // 32 bins = exactly one k-block of the output planes (bin0 = (m0 + 64 wr) / 2 is a multiple of 32), so per plane the wave
        // writes, for each of its 64 columns, one whole 64-byte row segment.  A lane holds PAIRS of bins (rows 8 g + 4 lh + {0..3} =
        // re, im, re, im), so the tile goes through the wave's 2-KB LDS tile, one 32-column half and plane at a time: written as
        // 4-byte words [column][32 bins] (16-byte chunks swizzled by (column >> 2) & 3), read back as 16-byte pieces, four lanes
        // per column -- 16 columns x 64 B per store instruction, a contiguous 1-KB run of the k-blocked planes.
        unsigned char* st = stage_small != nullptr ? stage_small : stage;
        unsigned short* Po = (unsigned short*)p.Pout;
        const int bins = p.Co >> 1;
        const int bin0 = (m0 + wr * 64) >> 1;
        const int lane = lr + 32 * lh;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            float q[2][4][2];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int bin = bin0 + 16 * ti + 4 * g + 2 * lh + e;
                        const float m = hypotf(acc[ti][tj][4 * g + 2 * e], acc[ti][tj][4 * g + 2 * e + 1]);
                        q[ti][g][e] = bin < bins ? m : 0.0f;
                    }
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // bins 16 ti + 4 g + 2 lh + {0, 1}: bytes 32 ti + 8 g + 4 lh of the column's 64
                        const unsigned h = pack_plane2<NP>(q[ti][g][0], q[ti][g][1]);
                        *(unsigned*)(st + lr * 64 + (((2 * ti + (g >> 1)) ^ ((lr >> 2) & 3)) << 4) + 8 * (g & 1) + 4 * lh) = h;
                        q[ti][g][0] -= __uint_as_float(h << 16);
                        q[ti][g][1] -= __uint_as_float(h & 0xffff0000u);
                    }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int idx = it * 64 + lane, cl = idx >> 2, chunk = idx & 3;
                    const u32x4 v = *(const u32x4*)(st + cl * 64 + ((chunk ^ ((cl >> 2) & 3)) << 4));
                    const int64_t col = c0 + wc * 64 + tj * 32 + cl;
                    const int b = bin0 + chunk * 8;
                    if (col < cols_pad && b < co_pad32) *(u32x4*)(Po + planes_at(pl, col, b, cols_pad, co_pad32)) = col < cols ? v : u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
        return;
    }
    // y_split (block-uniform): this tile's rows belong to the second output tensor
    const bool second = p.y_split > 0 && m0 >= p.y_split;
    float* const Yout = second ? p.Y2 : p.Y;
    const int row_off = second ? p.y_split : 0;
    const int co_out = p.y_split > 0 ? (second ? p.Co - p.y_split : p.y_split) : p.Co;
    unsigned obase[2];
    bool cok[2];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int64_t col = c0 + wc * 64 + tj * 32 + lr;
        cok[tj] = col < cols;
        const int64_t n = cok[tj] ? col / p.T : 0;
        obase[tj] = (unsigned)(n * co_out * p.T + (cok[tj] ? col - n * p.T : 0)) - (unsigned)(row_off * p.T);
    }
    const bool full_rows = m0 + GM <= p.Co;            // block-uniform
    // fp16 split planes: the operands were multiples s_W, s_P of the values -- the accumulator is brought back before the bias
    const float oscale = F16S ? p.wscale[0] * p.in_unscale : 1.0f;
    const float pscale = F16S ? p.pout_scale : 1.0f;
    const float* __restrict__ bias = p.bias;
    const float* __restrict__ post_add = p.post_add;
    const float* __restrict__ ch_scale = p.ch_scale;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int rbase = m0 + wr * 64 + ti * 32 + 4 * lh;
        float bia[4][4], pad[4][4], scl[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int row = rbase + 8 * g + e;
                row = row < p.Co ? row : p.Co - 1;
                bia[g][e] = bias != nullptr ? bias[row] : 0.0f;
                pad[g][e] = post_add != nullptr ? post_add[row] : 0.0f;
                scl[g][e] = ch_scale != nullptr ? ch_scale[row] : 1.0f;
            }
        float res[2][16];
        if (p.residual != nullptr) {
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = rbase + (r & 3) + 8 * (r >> 2);
                    row = row < p.Co ? row : p.Co - 1;
                    res[tj][r] = p.residual[obase[tj] + (unsigned)(row * p.T)];
                }
        }
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            float vv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = F16S ? fmaf(acc[ti][tj][r], oscale, bia[r >> 2][r & 3]) : acc[ti][tj][r] + bia[r >> 2][r & 3];
                if (ACT == 1) x = gelu_fast(x);
                else if (ACT == 2) x = expf(x);
                x = (x + pad[r >> 2][r & 3]) * scl[r >> 2][r & 3];
                if (p.residual != nullptr) x += res[tj][r];
                vv[r] = x;
            }
            if (Yout != nullptr && cok[tj]) {
                if (full_rows) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Yout[obase[tj] + (unsigned)((rbase + (r & 3) + 8 * (r >> 2)) * p.T)] = vv[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        if (row < p.Co) Yout[obase[tj] + (unsigned)(row * p.T)] = vv[r];
                    }
                }
            }
            if (p.Pout != nullptr && stage != nullptr) {
                // plane-packed output through the wave's LDS tile (columns past the end are written too, and skipped on the way out)
                const int cl = tj * 32 + lr;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int rowl = ti * 32 + 8 * g + 4 * lh;                 // channel inside the wave's 64
                    float q[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = (full_rows || rbase + 8 * g + e < p.Co) ? vv[4 * g + e] * pscale : 0.0f;
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        const unsigned h01 = split_step<NP, F16S>(q[0], q[1], cok[tj]), h23 = split_step<NP, F16S>(q[2], q[3], cok[tj]);
                        *(uint2*)(stage + pl * 8192 + cl * 128 + ((((rowl >> 3)) ^ (cl & 7)) << 4) + 8 * lh) = make_uint2(h01, h23);
                    }
                }
            } else if (p.Pout != nullptr && stage_small != nullptr) {
                // the same through a 2-KB wave-private tile, one 32 x 32 sub-tile and plane at a time (the persistent kernel: its
                // DMA ring is busy with the next tile during the epilogue, only 16 KB of LDS are free): rows of 64 B (32 channels),
                // chunk index XOR-ed with (column >> 2) & 3; leaves as 16-byte pieces, four lanes per column
                unsigned short* Po = (unsigned short*)p.Pout;
                float q[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[g][e] = (full_rows || rbase + 8 * g + e < p.Co) ? vv[4 * g + e] * pscale : 0.0f;
                const int lane = lr + 32 * lh;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const unsigned h01 = split_step<NP, F16S>(q[g][0], q[g][1], cok[tj]), h23 = split_step<NP, F16S>(q[g][2], q[g][3], cok[tj]);
                        *(uint2*)(stage_small + lr * 64 + ((g ^ ((lr >> 2) & 3)) << 4) + 8 * lh) = make_uint2(h01, h23);
                    }
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int idx = it * 64 + lane, cl = idx >> 2, chunk = idx & 3;
                        const u32x4 v = *(const u32x4*)(stage_small + cl * 64 + ((chunk ^ ((cl >> 2) & 3)) << 4));
                        const int64_t col = c0 + wc * 64 + tj * 32 + cl;
                        const int row0 = m0 + wr * 64 + ti * 32 + chunk * 8;
                        // (64 lanes: 16 columns x the 64 B of one k-block -- one contiguous 1-KB run of the k-blocked planes)
                        if (col < cols && row0 < co_pad32) *(u32x4*)(Po + planes_at(pl, col, row0, cols_pad, co_pad32)) = v;
                    }
                }
            } else if (p.Pout != nullptr && cok[tj]) {
                // plane-packed output for the next GEMM: 4 consecutive channels of one column -> 8 B per plane
                unsigned short* Po = (unsigned short*)p.Pout;
                const int64_t col = c0 + wc * 64 + tj * 32 + lr;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = rbase + 8 * g;
                    if (row >= co_pad32) continue;
                    float q[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = (full_rows || row + e < p.Co) ? vv[4 * g + e] * pscale : 0.0f;
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        const unsigned h01 = split_step<NP, F16S>(q[0], q[1]), h23 = split_step<NP, F16S>(q[2], q[3]);
                        *(uint2*)(Po + planes_at(pl, col, row, cols_pad, co_pad32)) = make_uint2(h01, h23);
                    }
                }
            }
        }
    }
    if (p.Pout != nullptr && stage != nullptr) {
        unsigned short* Po = (unsigned short*)p.Pout;
        const int lane = lr + 32 * lh;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            // one store instruction = 16 columns x the 64 B of one k-block: a contiguous 1-KB run of the k-blocked planes
            const int cl = (it & 3) * 16 + (lane >> 2), chunk = (it >> 2) * 4 + (lane & 3);
            const int64_t col = c0 + wc * 64 + cl;
            const int row0 = m0 + wr * 64 + chunk * 8;
            if (col >= cols || row0 >= co_pad32) continue;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const u32x4 v = *(const u32x4*)(stage + pl * 8192 + cl * 128 + ((chunk ^ (cl & 7)) << 4));
                *(u32x4*)(Po + planes_at(pl, col, row0, cols_pad, co_pad32)) = v;
            }
        }
    }
}

// KB2 (NP = 2 only; round 5): ONE fp16 plane per operand with a K-step of 64 -- the two "planes" of a stage are the two k-blocks of the
// step, the products are (block 0 x block 0) + (block 1 x block 1) on v_mfma_f32_32x32x16_f16, the output is one fp16 plane.  The plain
// (planes = 1) GEMM in the two-plane kernel's slot layout: 16 MFMAs between two block barriers instead of the 8 of a 32-deep step.
// F16S (NP = 2; round 5): the two planes are fp16 (hi, lo) of a power-of-two multiple of the values (AliveGemm.f16s): the same three
// products on v_mfma_f32_32x32x16_f16, the accumulator rescaled in the epilogue.
template <int NP, int NS, int MINB, int ACT, bool KB2 = false, bool F16S = false>
__global__ __launch_bounds__(256, MINB) void gemm_planes_kernel(AliveGemm p, int n_mt, int ntiles, int64_t cols,
                                                             int64_t cols_pad, int co_pad, int co_pad32, int kpad,
                                                             GemmWalk gw, long long* stamps) {
#ifdef ALIVE_STAMPS                 // diagnostic build only (make EXTRA=-DALIVE_STAMPS; tools/stamp_gemm.py)
    const long long ts0 = wall_clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SLOT = 2 * NP * PLANE_BYTES;
    constexpr int NI = 4 * NP;                    // DMA pieces per wave per step
    constexpr int NPROD = KB2 ? 2 : NP * (NP + 1) / 2;
    static_assert(!KB2 || NP == 2, "the 64-deep one-plane form lives in the two-plane kernel's slots");
    static_assert(!F16S || (NP == 2 && !KB2), "fp16 split planes: two planes, 32-deep steps");

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int lr = lane & 31, lh = lane >> 5;

    // XCD-aware tile order: the blocks an XCD runs together share column tiles (and the L2 copy of their X rows)
    int v;
    {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3, q = ntiles >> 3, r = ntiles & 7;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    int mt, ct;
    tile_of(v, n_mt, gw, mt, ct);
    const int m0 = mt * GM;
    const int64_t c0 = (int64_t)ct * GN;
    const int nsteps = kpad / (KB2 ? 2 * GK : GK);

    // ---- DMA geometry: piece q = w + 4 i  ->  (operand, plane, 16-row group) ----
    const int prow = lane >> 2;
    const int pchunk = (lane & 3) ^ ((prow >> 2) & 3);
    const unsigned short* src[NI];
    int ldst[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = w + 4 * i;
        const int op = q / (8 * NP), pl = (q % (8 * NP)) / 8, g = q % 8;
        const int spl = KB2 ? 0 : pl;                    // KB2: slot "plane" 1 is the step's second k-block of the ONE plane (walk1 below)
        const int r = g * 16 + prow;
        if (op == 0) {
            int row = m0 + r;
            row = row < co_pad ? row : co_pad - 1;
            src[i] = (const unsigned short*)p.W + planes_at(spl, row, 0, co_pad, kpad) + pchunk * 8;
        } else if (p.b_row == 0) {
            int64_t col = c0 + r;
            col = col < cols_pad ? col : cols_pad - 1;
            src[i] = (const unsigned short*)p.P + planes_at(spl, col, 0, cols_pad, kpad) + pchunk * 8;
        } else {                                         // custom row placement (overlapping rows: the STFT)
            int64_t col = c0 + r;
            col = col < cols ? col : cols - 1;
            const int64_t nn = col / p.T;
            src[i] = (const unsigned short*)p.P + (size_t)spl * p.b_plane + (size_t)nn * p.b_win + (size_t)(col - nn * p.T) * p.b_row + pchunk * 8;
        }
        ldst[i] = (op * NP + pl) * PLANE_BYTES + g * 1024;
    }
    // the steps are issued in order: `walk` holds the offsets of the next step to issue (pieces i < 2 NP are W's, the rest B's)
    StepWalk walk, walk1;                                // walk1 (KB2): one k-block further -- the slot's second half
    if constexpr (KB2) walk1.advance(gw);
    auto issue = [&](int step, int i) {
        // (q = w + 4 i with w < 4: the plane of piece i is (i >> 1) & 1 at NP = 2)
        const StepWalk& wk = (KB2 && ((i >> 1) & 1)) ? walk1 : walk;
        __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + (i < 2 * NP ? wk.a : wk.b)), (lptr_t)(smem + (step % NS) * SLOT + ldst[i]), 16, 0, 0);
    };
    auto next_step = [&]() {
        walk.advance(gw);
        if constexpr (KB2) { walk.advance(gw); walk1.advance(gw); walk1.advance(gw); }
    };

    // the bias joins in the epilogue: nothing but the DMA issue stands between the launch and the first MFMA
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.0f; acc[i][1][r] = 0.0f; }

#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < nsteps) {
#pragma unroll
            for (int i = 0; i < NI; ++i) issue(s, i);
            next_step();
        }

    // fragment byte offsets inside a plane: row * 64 + ((2 ks + lh) ^ ((row >> 2) & 3)) * 16 ; ks = 1 flips bit 5
    const int sw = (lr >> 2) & 3;
    const int a_off = (wr * 64 + lr) * 64 + ((lh ^ sw) << 4);
    const int b_off = NP * PLANE_BYTES + (wc * 64 + lr) * 64 + ((lh ^ sw) << 4);

    bf16x8 fa[2][2][NP], fb[2][2][NP];            // [ks parity][tile][plane]
    auto load_frags = [&](int step, int ks, bf16x8 (&a)[2][NP], bf16x8 (&b)[2][NP]) {
        const unsigned char* S = smem + (step % NS) * SLOT;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                a[t][pl] = *(const bf16x8*)(S + ((a_off + t * 2048) ^ (ks << 5)) + pl * PLANE_BYTES);
                b[t][pl] = *(const bf16x8*)(S + ((b_off + t * 2048) ^ (ks << 5)) + pl * PLANE_BYTES);
            }
    };
    // all plane products (i, j), i + j <= NP - 1, smallest terms first; the four accumulator tiles are interleaved so
    // that dependent MFMAs are four issues apart.  `hook(n)` runs after the n-th group of four (DMA issue slots).
    auto mma = [&](bf16x8 (&a)[2][NP], bf16x8 (&b)[2][NP], auto&& hook) {
        int n = 0;
        if constexpr (KB2) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = mfma_f16(a[ti][kb], b[tj][kb], acc[ti][tj]);
                hook(n++);
            }
            return;
        }
#pragma unroll
        for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
            for (int i = 0; i <= sum; ++i) {
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
                        acc[ti][tj] = (NP == 1 || F16S) ? mfma_f16(a[ti][i], b[tj][sum - i], acc[ti][tj])
                                                        : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][i], b[tj][sum - i], acc[ti][tj], 0, 0, 0);
                hook(n++);
            }
    };

#ifdef ALIVE_STAMPS
    const long long ts1 = wall_clock64();
#endif
    // stage 0 landed?  (NS - 1 younger stages may still be in flight)
    if (nsteps >= NS) wait_vmcnt<(NS - 1) * NI>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, fa[0], fb[0]);
#ifdef ALIVE_STAMPS
    const long long ts2 = wall_clock64();
    long long cy_vm = 0, cy_bar = 0;
    const long long cy_l0 = __builtin_readcyclecounter();
#endif

    for (int s = 0; s < nsteps; ++s) {
        // the second half's fragments are requested behind the first MFMA group (a request in front of it would be
        // waited for together with the first half's: lgkmcnt is in order and the loop edge hides the count)
        mma(fa[0], fb[0], [&](int n) {
            if (n == 0) load_frags(s, 1, fa[1], fb[1]);
        });
        __builtin_amdgcn_sched_barrier(0);
        // step s + 1 must have landed for every wave; steps s + 2 .. s + NS - 1 stay in flight
        wait_lgkmcnt0();
#ifdef ALIVE_STAMPS
        const long long cy0 = __builtin_readcyclecounter();
#endif
        if (s + NS - 1 < nsteps) wait_vmcnt<(NS - 2) * NI>(); else wait_vmcnt<0>();
#ifdef ALIVE_STAMPS
        const long long cy1 = __builtin_readcyclecounter();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef ALIVE_STAMPS
        const long long cy2 = __builtin_readcyclecounter();
        cy_vm += cy1 - cy0; cy_bar += cy2 - cy1;
#endif
        if (s + 1 < nsteps) load_frags(s + 1, 0, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        // slot of step s is free now: refill it with step s + NS, NI pieces spread over the NPROD MFMA groups
        const bool refill = s + NS < nsteps;
        mma(fa[1], fb[1], [&](int n) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (i * NPROD / NI == n && refill) issue(s + NS, i);
        });
        next_step();
        __builtin_amdgcn_sched_barrier(0);
    }
    wait_vmcnt<0>();
#ifdef ALIVE_STAMPS
    const long long ts3 = wall_clock64();
    const long long cy_l1 = __builtin_readcyclecounter();
#endif

    unsigned char* stage = nullptr;
    if constexpr (ACT != 3) {
        if (p.Pout != nullptr) {                       // block-uniform: the DMA ring is free now; each wave takes NP x 8 KB of it
            __syncthreads();                           // every wave has read its last fragments
            stage = smem + w * (NP * 8192);
        }
    }
    gemm_epilogue<KB2 ? 1 : NP, ACT, F16S>(p, acc, m0, c0, wr, wc, lr, lh, cols, cols_pad, co_pad32, stage);
#ifdef ALIVE_STAMPS
    if (stamps != nullptr && tid == 0) {
        long long* o = stamps + (size_t)blockIdx.x * 8;
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3; o[4] = wall_clock64();
        o[5] = cy_l1 - cy_l0;          // core-clock cycles of the k-loop, of which waiting for the DMA stage / at the barrier:
        o[6] = cy_vm; o[7] = cy_bar;
    }
#endif
}

// ---- persistent form with LOADER WAVES: one block per CU walks its tiles, the DMA ring runs ACROSS the tile seams ----
// In the one-tile kernel a block spends 3 us issuing its first stages and 7 us in its epilogue next to 12 - 22 us of k-steps
// (in-kernel stamps), with nothing else resident on the CU at NP = 3.  Here the refill of step s targets step
// s + NS of the block's tile STREAM: once a tile's last steps are reached the pieces come from the next tile, so that tile's
// first NS stages land under this tile's epilogue and the next main loop starts on landed data.  Needs nsteps >= NS.
// Since round 4 the block has 8 waves: 0 - 3 compute, 4 - 7 only issue the LDS-DMA.  An LDS-DMA instruction holds the wave that
// issues it for 60 - 180 cycles (MI355X_MICROARCH.md) and a wave issues in order, so in the four-wave form (rounds 2 - 3) the next
// MFMA of the issuing wave waited too while the matrix pipe drained: cycle stamps of that k-loop showed 2.6 % of it waiting for a
// DMA stage, 2.1 % at the barrier -- and 51 cycles per MFMA instead of 32.  A loader wave on the same SIMD stalls for free; the
// compute waves carry fragment reads and MFMAs only (no source pointers, no vmcnt: 235 instead of 338 registers), and their epilogue
// loads / stores no longer share a counter with the ring.  One block barrier per K-step as before: the loaders arrive after their
// counted wait for stage s + 1, the compute waves after their last fragment read of step s; behind it slot s is refilled with
// stream element s + NS.  Against the four-wave form, same tile order: 3 - 13 % per layer (tools/experiments/README.md).
template <int NP, int NS, int ACT>
__global__ __launch_bounds__(512, 1) void gemm_planes_lw_kernel(AliveGemm p, int n_mt, int ntiles, int64_t cols, int64_t cols_pad,
                                                                int co_pad, int co_pad32, int kpad, GemmWalk gw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SLOT = 2 * NP * PLANE_BYTES;
    constexpr int NI = 4 * NP;
    [[maybe_unused]] constexpr int NPROD = NP * (NP + 1) / 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsteps = kpad / GK;

    int v, v_end;
    const int v_stride = gridDim.x >> 3;
    {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, q = ntiles >> 3, r = ntiles & 7;
        const int beg = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        v_end = beg + q + (xcd < r ? 1 : 0);
        v = beg + j;
    }
    if (v >= v_end) return;

    if (w8 >= 4) {
        // ================= loader wave =================
        const int w = w8 - 4;
        const int prow = lane >> 2;
        const int pchunk = (lane & 3) ^ ((prow >> 2) & 3);
        int ldst[NI];
        unsigned off[2][NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = w + 4 * i;
            ldst[i] = ((i >= 2 * NP ? NP : 0) + (q % (8 * NP)) / 8) * PLANE_BYTES + (q % 8) * 1024;
        }
        auto tile_offsets = [&](int tv, unsigned (&o)[NI]) {
#ifdef ALIVE_LW_ABL_SAMETILE
            tv = (blockIdx.x & 7) + 8 * (tv & 3);          // timing only: 4 tiles per XCD, everything hits L2
#endif
            int mt, ct;
            tile_of(tv, n_mt, gw, mt, ct);
            const int64_t c0 = (int64_t)ct * GN;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int q = w + 4 * i;
                const int pl = (q % (8 * NP)) / 8, r = (q % 8) * 16 + prow;
                if (i < 2 * NP) {
                    int row = mt * GM + r;
                    row = row < co_pad ? row : co_pad - 1;
                    o[i] = (unsigned)((planes_at(pl, row, 0, co_pad, kpad) + pchunk * 8) * 2);
                } else if (p.b_row == 0) {
                    int64_t col = c0 + r;
                    col = col < cols_pad ? col : cols_pad - 1;
                    o[i] = (unsigned)((planes_at(pl, col, 0, cols_pad, kpad) + pchunk * 8) * 2);
                } else {
                    int64_t col = c0 + r;
                    col = col < cols ? col : cols - 1;
                    const int64_t nn = col / p.T;
                    o[i] = (unsigned)(((size_t)pl * p.b_plane + (size_t)nn * p.b_win + (size_t)(col - nn * p.T) * p.b_row + pchunk * 8) * 2);
                }
            }
        };
        StepWalk walk;
        auto issue = [&](int which, int slot, int i) {
            const unsigned char* base = i >= 2 * NP ? (const unsigned char*)p.P : (const unsigned char*)p.W;
            __builtin_amdgcn_global_load_lds((gptr_t)(base + off[which][i] + (unsigned)((i >= 2 * NP ? walk.b : walk.a) * 2)),
                                             (lptr_t)(smem + slot * SLOT + ldst[i]), 16, 0, 0);
        };
        int cur = 0, qb = 0;
        tile_offsets(v, off[0]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < NI; ++i) issue(0, s, i);
            walk.advance(gw);
        }
        wait_vmcnt<(NS - 1) * NI>();             // stage 0 of the first tile (this wave's pieces)
        __builtin_amdgcn_s_barrier();            // ... is visible to the compute waves
        while (true) {
            const int vn = v + v_stride;
            const bool has_next = vn < v_end;
            if (has_next) tile_offsets(vn, off[cur ^ 1]);
            for (int s = 0; s < nsteps; ++s) {
                const int slot = (qb + s) % NS;
                if (s + NS - 1 < nsteps || has_next) wait_vmcnt<(NS - 2) * NI>(); else wait_vmcnt<0>();      // stage s + 1 has landed
                __builtin_amdgcn_s_barrier();    // and every compute wave has read the last fragment of step s
                const int ts = s + NS;
                const bool in_tile = ts < nsteps;
                if (ts == nsteps) walk.reset();
#ifndef ALIVE_LW_ABL_NODMA
                if (in_tile || has_next) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) issue(in_tile ? cur : (cur ^ 1), slot, i);
                }
#endif
                walk.advance(gw);
            }
            if (!has_next) break;
            v = vn;
            cur ^= 1;
            qb = (qb + nsteps) % NS;
        }
        return;
    }

    // ================= compute wave =================
    const int w = w8;
    const int wr = w >> 1, wc = w & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 2) & 3;
    const int a_off = (wr * 64 + lr) * 64 + ((lh ^ sw) << 4);
    const int b_off = NP * PLANE_BYTES + (wc * 64 + lr) * 64 + ((lh ^ sw) << 4);
    bf16x8 fa[2][2][NP], fb[2][2][NP];
    auto load_frags = [&](int slot, int ks, bf16x8 (&a)[2][NP], bf16x8 (&b)[2][NP]) {
        const unsigned char* S = smem + slot * SLOT;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                a[t][pl] = *(const bf16x8*)(S + ((a_off + t * 2048) ^ (ks << 5)) + pl * PLANE_BYTES);
                b[t][pl] = *(const bf16x8*)(S + ((b_off + t * 2048) ^ (ks << 5)) + pl * PLANE_BYTES);
            }
    };
    f32x16 acc[2][2];
    auto mma = [&](bf16x8 (&a)[2][NP], bf16x8 (&b)[2][NP], auto&& hook) {
        int n = 0;
#pragma unroll
        for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
            for (int i = 0; i <= sum; ++i) {
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
                        acc[ti][tj] = NP == 1 ? mfma_f16(a[ti][i], b[tj][sum - i], acc[ti][tj])
                                              : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][i], b[tj][sum - i], acc[ti][tj], 0, 0, 0);
                hook(n++);
            }
    };

    int qb = 0;
    __builtin_amdgcn_s_barrier();                // stage 0 of the first tile is visible
    while (true) {
        const int vn = v + v_stride;
        const bool has_next = vn < v_end;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.0f; acc[i][1][r] = 0.0f; }
        load_frags(qb, 0, fa[0], fb[0]);         // (visible since the last barrier of the tile before)
        for (int s = 0; s < nsteps; ++s) {
            const int slot = (qb + s) % NS;
            const int slot1 = slot + 1 == NS ? 0 : slot + 1;
            mma(fa[0], fb[0], [&](int n) {
#ifdef ALIVE_LW_ABL_NOREAD
                if (n == 0 && s == 0) load_frags(slot, 1, fa[1], fb[1]);
#else
                if (n == 0) load_frags(slot, 1, fa[1], fb[1]);
#endif
            });
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkmcnt0();
            __builtin_amdgcn_s_barrier();
#ifndef ALIVE_LW_ABL_NOREAD
            if (s + 1 < nsteps) load_frags(slot1, 0, fa[0], fb[0]);
#endif
            __builtin_amdgcn_sched_barrier(0);
            mma(fa[1], fb[1], [&](int) {});
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef ALIVE_LW_ABL_NOEPI
        if (nsteps < 0)
#endif
        {
            int mt, ct;
            tile_of(v, n_mt, gw, mt, ct);
            gemm_epilogue<NP, ACT>(p, acc, mt * GM, (int64_t)ct * GN, wr, wc, lr, lh, cols, cols_pad, co_pad32, nullptr,
                                   p.Pout != nullptr ? smem + NS * SLOT + w * 2048 : nullptr);
        }
#ifdef ALIVE_LW_ABL_NOEPI
        else {
#pragma unroll
            for (int i = 0; i < 2; ++i) { asm volatile("" :: "v"(acc[i][0])); asm volatile("" :: "v"(acc[i][1])); }
        }
#endif
        if (!has_next) break;
        v = vn;
        qb = (qb + nsteps) % NS;
    }
}

// fp32 [N][C][T] -> k-blocked planes [NP][C_pad / 32][cols_pad][32] (zero padded in both directions)
template <int NP>
__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ X, int C, int T, int64_t cols, int64_t cols_pad,
                                                        int c_pad, unsigned short* __restrict__ P) {
    __shared__ float tile[64][65];                 // [channel][column]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t col0 = (int64_t)blockIdx.x * 64;
    const int ch0 = blockIdx.y * 64;
    {
        const int64_t col = col0 + lane;
        const bool ok = col < cols;
        const int64_t n = ok ? col / T : 0;
        const int t = ok ? (int)(col - n * T) : 0;
        const float* xc = X + (size_t)n * C * T + t;
        for (int r = wv; r < 64; r += 4) {
            const int c = ch0 + r;
            tile[r][lane] = (ok && c < C) ? xc[(size_t)c * T] : 0.0f;
        }
    }
    __syncthreads();
    // thread -> (k-block half of the tile, column, 8-channel chunk inside the block): 2 x 64 x 4 = 512 items, two per thread; a wave
    // writes 16 columns x 64 B = one contiguous 1-KB run per plane
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = it * 256 + threadIdx.x;
        const int ck = it * 4 + (item & 3), cl = (item >> 2) & 63;
        const int64_t col = col0 + cl;
        const int c = ch0 + ck * 8;
        if (col >= cols_pad || c >= c_pad) continue;
        float vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = tile[ck * 8 + e][cl];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned h = pack_plane2<NP>(vv[2 * e], vv[2 * e + 1]);
                o[e] = h;
                vv[2 * e] -= __uint_as_float(h << 16);
                vv[2 * e + 1] -= __uint_as_float(h & 0xffff0000u);
            }
            *(u32x4*)(P + planes_at(pl, col, c, cols_pad, c_pad)) = o;
        }
    }
}

long long* g_stamps = nullptr;
inline int64_t pad_cols(int64_t cols) { return (cols + GN - 1) / GN * GN; }
inline int pad32(int c) { return (c + 31) & ~31; }
inline GemmWalk make_walk(const AliveGemm& d) {
    GemmWalk g;
    g.a_ks = (int64_t)((d.Co + 15) & ~15) * GK;
    if (d.b_row == 0) { g.b_blk = pad_cols((int64_t)d.N * d.T) * GK; g.b_tap = 0; g.ncb = pad32(d.Ci) / GK; }
    else if (d.b_cblk == 0) { g.b_blk = 0; g.b_tap = GK; g.ncb = 1; }
    else { g.b_blk = d.b_blk; g.b_tap = GK; g.ncb = d.b_cblk; }
    // row tiles per group: their weight tiles (128 rows x K x planes) should fit in about 2.5 MB of an XCD's L2; even groups
    static const int rg_env = getenv("ALIVE_GEMM_RG") ? atoi(getenv("ALIVE_GEMM_RG")) : 0;
    const int n_mt = cdiv(d.Co, GM);
    const int64_t tile_bytes = (int64_t)GM * pad32(d.Ci) * d.planes * 2;
    int rg = rg_env > 0 ? rg_env : (int)std::max<int64_t>(1, (int64_t)(2.5 * 1048576) / tile_bytes);
    rg = std::min(rg, n_mt);
    const int n_rg = cdiv(n_mt, rg);
    g.rg = cdiv(n_mt, n_rg);
    g.n_ct = cdiv((int64_t)d.N * d.T, GN);
    return g;
}

template <int NP, int NS, int MINB, int ACT, bool KB2 = false, bool F16S = false>
int launch_gemm_act(const AliveGemm& d, hipStream_t s) {
    constexpr int LDS = NS * 2 * NP * PLANE_BYTES;
    {
        static LdsOptIn optin;
        hipError_t e = optin.ensure({(const void*)gemm_planes_kernel<NP, NS, MINB, ACT, KB2, F16S>}, LDS);
        if (e != hipSuccess) {
            alive_set_error("alive_gemm_planes: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return ALIVE_ERR_LAUNCH;
        }
    }
    const int64_t cols = (int64_t)d.N * d.T;
    const int n_mt = cdiv(d.Co, GM), n_ct = cdiv(cols, GN);
    const int ntiles = n_mt * n_ct;
    gemm_planes_kernel<NP, NS, MINB, ACT, KB2, F16S><<<ntiles, 256, LDS, s>>>(
        d, n_mt, ntiles, cols, pad_cols(cols), (d.Co + 15) & ~15, pad32(ACT == 4 ? d.Co / 2 : d.Co), pad32(d.Ci), make_walk(d),
        g_stamps);
    ALIVE_CHECK_LAUNCH("alive_gemm_planes");
    return ALIVE_OK;
}

template <int NP, int NS, int ACT>
int launch_gemm_lw_act(const AliveGemm& d, hipStream_t s) {
    constexpr int LDS = NS * 2 * NP * PLANE_BYTES + 4 * 2048;
    {
        static LdsOptIn optin;
        hipError_t e = optin.ensure({(const void*)gemm_planes_lw_kernel<NP, NS, ACT>}, LDS);
        if (e != hipSuccess) {
            alive_set_error("alive_gemm_planes: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return ALIVE_ERR_LAUNCH;
        }
    }
    const int64_t cols = (int64_t)d.N * d.T;
    const int n_mt = cdiv(d.Co, GM), n_ct = cdiv(cols, GN);
    gemm_planes_lw_kernel<NP, NS, ACT><<<256, 512, LDS, s>>>(d, n_mt, n_mt * n_ct, cols, pad_cols(cols), (d.Co + 15) & ~15,
                                                            pad32(ACT == 4 ? d.Co / 2 : d.Co), pad32(d.Ci), make_walk(d));
    ALIVE_CHECK_LAUNCH("alive_gemm_planes(persistent)");
    return ALIVE_OK;
}

template <int NP, int NS>
int launch_gemm_lw(const AliveGemm& d, hipStream_t s) {
    if constexpr (NP == 3) { if (d.act == 3) return launch_gemm_lw_act<NP, NS, 3>(d, s); }
    if constexpr (NP == 3) { if (d.act == 4) return launch_gemm_lw_act<NP, NS, 4>(d, s); }
    if (d.act == 1) return launch_gemm_lw_act<NP, NS, 1>(d, s);
    if (d.act == 2) return launch_gemm_lw_act<NP, NS, 2>(d, s);
    return launch_gemm_lw_act<NP, NS, 0>(d, s);
}

template <int NP, int NS, int MINB>
int launch_gemm(const AliveGemm& d, hipStream_t s) {
    if constexpr (NP == 3) { if (d.act == 3) return launch_gemm_act<NP, NS, MINB, 3>(d, s); }
    if constexpr (NP == 3) { if (d.act == 4) return launch_gemm_act<NP, NS, MINB, 4>(d, s); }
    if (d.act == 1) return launch_gemm_act<NP, NS, MINB, 1>(d, s);
    if (d.act == 2) return launch_gemm_act<NP, NS, MINB, 2>(d, s);
    return launch_gemm_act<NP, NS, MINB, 0>(d, s);
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void argmax_merge_kernel(const float* __restrict__ val, const int* __restrict__ idx, int nblk,
                                                           int64_t cols, float* __restrict__ out) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    float b = val[col];
    int i = idx[col];
    for (int k = 1; k < nblk; ++k) {
        const float v = val[(size_t)k * cols + col];
        const int vi = idx[(size_t)k * cols + col];
        if ((v > b) || (v == b && vi < i) || (v != v && b == b)) { b = v; i = vi; }
    }
    out[col] = (float)i;
}
}  // namespace

extern "C" int alive_argmax_merge(const float* arg_val, const int32_t* arg_idx, int nblk, int64_t cols, float* out, void* stream) {
    ALIVE_CHECK_ARG(arg_val && arg_idx && out && nblk > 0 && cols > 0, "alive_argmax_merge: bad args");
    argmax_merge_kernel<<<(unsigned)((cols + 255) / 256), 256, 0, (hipStream_t)stream>>>(arg_val, arg_idx, nblk, cols, out);
    ALIVE_CHECK_LAUNCH("alive_argmax_merge");
    return ALIVE_OK;
}

extern "C" void alive_debug_set_stamps(long long* p) { g_stamps = p; }

extern "C" size_t alive_planes_bytes(int64_t cols, int C, int planes) {
    return (size_t)planes * (size_t)pad_cols(cols) * pad32(C) * 2;
}

extern "C" int alive_to_planes(const float* X, int N, int C, int T, int planes, void* P, void* stream) {
    ALIVE_CHECK_ARG(X && P && N > 0 && C > 0 && T > 0, "alive_to_planes: bad args");
    ALIVE_CHECK_ARG(planes >= 1 && planes <= 3, "alive_to_planes: planes must be 1 (one fp16 plane), 2 or 3 (split bf16), got %d", planes);
    ALIVE_CHECK_ARG((((uintptr_t)P) & 15) == 0, "alive_to_planes: P must be 16-byte aligned");
    const int64_t cols = (int64_t)N * T, cp = pad_cols(cols);
    const int c_pad = pad32(C);
    dim3 g((unsigned)(cp / 64), cdiv(c_pad, 64));
    if (planes == 1) to_planes_kernel<1><<<g, 256, 0, (hipStream_t)stream>>>(X, C, T, cols, cp, c_pad, (unsigned short*)P);
    else if (planes == 2) to_planes_kernel<2><<<g, 256, 0, (hipStream_t)stream>>>(X, C, T, cols, cp, c_pad, (unsigned short*)P);
    else to_planes_kernel<3><<<g, 256, 0, (hipStream_t)stream>>>(X, C, T, cols, cp, c_pad, (unsigned short*)P);
    ALIVE_CHECK_LAUNCH("alive_to_planes");
    return ALIVE_OK;
}

extern "C" int alive_gemm_planes(const AliveGemm* d, void* stream) {
    ALIVE_CHECK_ARG(d && d->W && d->P, "alive_gemm_planes: null pointer");
    if (d->act == 3) {
        ALIVE_CHECK_ARG((d->planes == 3 || (d->planes == 2 && d->f16s)) && d->arg_val && d->arg_idx && !d->Y && !d->Pout && !d->residual &&
                        !d->post_add && !d->ch_scale,
                        "alive_gemm_planes: act 3 (argmax) takes 3 planes (or fp16 split planes), arg_val / arg_idx and no other output or epilogue term");
    } else if (d->act == 4) {
        ALIVE_CHECK_ARG(d->planes == 3 && d->Pout && !d->Y && !d->bias && !d->residual && !d->post_add && !d->ch_scale && !d->y_split && (d->Co & 1) == 0,
                        "alive_gemm_planes: act 4 (magnitude of row pairs) takes 3 planes, an even Co, Pout and no other output or epilogue term");
    } else {
        ALIVE_CHECK_ARG(d->Y || d->Pout, "alive_gemm_planes: no output");
    }
    ALIVE_CHECK_ARG(d->y_split == 0 || (d->y_split % GM == 0 && d->y_split < d->Co && d->Y && d->Y2 && !d->residual && !d->Pout && d->act != 3),
                    "alive_gemm_planes: y_split must be a multiple of %d below Co, with Y and Y2 and neither residual nor Pout", GM);
    ALIVE_CHECK_ARG(d->N > 0 && d->T > 0 && d->Ci > 0 && d->Co > 0, "alive_gemm_planes: bad shape");
    ALIVE_CHECK_ARG(d->planes >= 1 && d->planes <= 3, "alive_gemm_planes: planes must be 1, 2 or 3, got %d", d->planes);
    ALIVE_CHECK_ARG(d->planes != 1 || (d->act != 3 && d->act != 4), "alive_gemm_planes: one plane (plain fp16) has no argmax / magnitude epilogue");
    ALIVE_CHECK_ARG(d->b_row == 0 || ((d->b_row | d->b_win | d->b_plane) & 7) == 0, "alive_gemm_planes: custom row placement must be in multiples of 8 elements");
    ALIVE_CHECK_ARG(d->b_row == 0 || (d->Ci & 31) == 0, "alive_gemm_planes: custom row placement needs Ci %% 32 == 0");
    ALIVE_CHECK_ARG(d->b_cblk >= 0 && (d->b_cblk == 0 || (d->b_row != 0 && (d->Ci / 32) % d->b_cblk == 0 && (d->b_blk & 7) == 0)),
                    "alive_gemm_planes: b_cblk %d (k-blocks per tap) must divide Ci / 32 = %d, with b_row and b_blk set", d->b_cblk, d->Ci / 32);
    ALIVE_CHECK_ARG(d->act >= 0 && d->act <= 4, "alive_gemm_planes: activation %d", d->act);
    ALIVE_CHECK_ARG(((((uintptr_t)d->W) | ((uintptr_t)d->P) | ((uintptr_t)d->Pout)) & 15) == 0,
                    "alive_gemm_planes: W / P / Pout must be 16-byte aligned");
    ALIVE_CHECK_ARG(!(d->Y || d->residual) || (int64_t)d->N * d->Co * d->T < (1ll << 30),
                    "alive_gemm_planes: fp32 tensor of %lld elements exceeds the 32-bit offsets", (long long)d->N * d->Co * d->T);
    // variants (A/B switch for tools/bench_gemm_planes.py): 0 = default, 1 = one-tile kernels only, 2 = persistent for both
    static const int variant = getenv("ALIVE_GEMM_VARIANT") ? atoi(getenv("ALIVE_GEMM_VARIANT")) : 0;
    const int64_t ntiles = (int64_t)cdiv(d->Co, GM) * cdiv((int64_t)d->N * d->T, GN);
    const int nsteps = pad32(d->Ci) / GK;
    // from 2 tiles per CU on (512; 1024 until the loader waves: with them the form pays below four tiles per CU too -- 512 -> 256 x 3
    // planes, 900 tiles: 0.125 -> 0.116 ms)
    static const int persist_min = getenv("ALIVE_GEMM_PERSIST_MIN") ? atoi(getenv("ALIVE_GEMM_PERSIST_MIN")) : 512;
    const bool can_persist = ntiles >= persist_min && variant != 1 &&
                             (d->b_row == 0 ? (int64_t)d->planes * pad_cols((int64_t)d->N * d->T) * pad32(d->Ci)
                                            : (int64_t)d->planes * d->b_plane) * 2 < (1ll << 32);          // 32-bit DMA offsets
    if (d->planes == 1) {
        // one plane = plain fp16 operands, one MFMA per product (round 5): W and P are single fp16 planes in the k-blocked layout.
        // Stages of 16 KB, four in the ring, two blocks per CU.
        static const int form1 = getenv("ALIVE_GEMM1_FORM") ? atoi(getenv("ALIVE_GEMM1_FORM")) : 0;      // A/B: ring depth x blocks per CU
        // K a multiple of 64: the 64-deep step in the two-plane kernel's slots (form 4 forces the 32-deep kernel)
        if ((pad32(d->Ci) & 63) == 0 && pad32(d->Ci) >= 128 && form1 == 0) {
            if (d->act == 1) return launch_gemm_act<2, 2, 2, 1, true>(*d, (hipStream_t)stream);
            if (d->act == 2) return launch_gemm_act<2, 2, 2, 2, true>(*d, (hipStream_t)stream);
            return launch_gemm_act<2, 2, 2, 0, true>(*d, (hipStream_t)stream);
        }
        if (form1 == 1) return launch_gemm<1, 3, 3>(*d, (hipStream_t)stream);
        if (form1 == 2) return launch_gemm<1, 6, 2>(*d, (hipStream_t)stream);
        if (form1 == 3) return launch_gemm<1, 2, 4>(*d, (hipStream_t)stream);
        return launch_gemm<1, 4, 2>(*d, (hipStream_t)stream);
    }
    ALIVE_CHECK_ARG(!d->f16s || (d->planes == 2 && d->wscale != nullptr && d->in_unscale > 0.0f && d->act >= 0 && d->act <= 3 &&
                                 (d->Pout == nullptr || d->pout_scale > 0.0f) && d->y_split == 0),
                    "alive_gemm_planes: f16s (fp16 split planes) takes planes = 2, wscale, in_unscale > 0, act 0 - 3 and pout_scale > 0 with Pout");
    if (d->planes == 2 && d->f16s) {
        if (d->act == 3) return launch_gemm_act<2, 2, 2, 3, false, true>(*d, (hipStream_t)stream);
        if (d->act == 1) return launch_gemm_act<2, 2, 2, 1, false, true>(*d, (hipStream_t)stream);
        if (d->act == 2) return launch_gemm_act<2, 2, 2, 2, false, true>(*d, (hipStream_t)stream);
        return launch_gemm_act<2, 2, 2, 0, false, true>(*d, (hipStream_t)stream);
    }
    if (d->planes == 2) {
        // two planes: the one-tile kernel with two blocks per CU (one block's epilogue under the other's MFMAs) beats one persistent block
        if (can_persist && variant == 2 && nsteps >= 4) return launch_gemm_lw<2, 4>(*d, (hipStream_t)stream);
        return launch_gemm<2, 2, 2>(*d, (hipStream_t)stream);
    }
    if (can_persist && nsteps >= 3) return launch_gemm_lw<3, 3>(*d, (hipStream_t)stream);
    return launch_gemm<3, 3, 1>(*d, (hipStream_t)stream);
}

ALIVE_F16_SAT_GETTER(alive_f16_sat_gemm)
