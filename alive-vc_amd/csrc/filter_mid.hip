// FilterBlock.forward for the 64-channel scale of the decoder's U-Net (/root/reference/module/decoder.py:105-150),
// fused into ONE kernel like the 16- / 8-channel scales (filter_small.hip), on the split-bf16 MFMA.
//
// Conv by conv this scale (36 000 samples x 64 channels per window) is HBM-bound: every k5 conv reads and writes one or
// two full fp32 tensors (4.7 GB per conv at 128 windows against 0.5 ms of MFMA work).  Here a block owns a tile of
// 256 columns (200 outputs + the 56-column causal halo 4 x (1+1+2+2+4+4), recomputed instead of exchanged) and keeps
// it on chip through the 1x1 input conv and the six GELU -> FiLM -> reflect-left causal k5 convs:
//   * the modulated conv input z and the intermediate y live in two LDS buffers, ALREADY SPLIT into two bf16 planes
//     (hi = bf16(v), lo = bf16(v - hi)) and stored [column][64 channels] so that an MFMA B fragment is one ds_read_b128;
//     the 16-B chunk order of a 128-B row is XOR-swizzled with (row & 7): fragment reads are conflict-free at any tap
//     shift;
//   * a wave owns 16 output channels for all 256 columns (v_mfma_f32_16x16x32_bf16): its A fragments (10 k-steps x 2
//     planes = 80 VGPRs) are loaded from L2 once per conv and stay in registers, the residual stream h (16 column tiles
//     x 4) lives in registers in the MFMA C layout, so residual add, GELU, FiLM and the re-split of a tile are
//     register-local and only the modulated planes go back to LDS;
//   * a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, smallest terms first, fp32 accumulate (the "bf16x3" product of
//     conv_split.hip, ~2^-16 per product).
// HBM traffic drops to the input, the U-Net skip and the output (3 tensor passes instead of ~18).
#include "conv_epilogue.h"

// With hipcc's SLP vectorizer on, filter_block64_kernel comes out as a binary whose last column tiles differ from run to run
// (DESIGN.md 3.2b'; a static scan of that binary, tools/mfma_hazard_scan.py, finds no MFMA operand / result distance below
// the gfx950 tables, so the defect is not one the listed software hazards explain).  The Makefile compiles this file with
// -fno-slp-vectorize and says so with the macro; any other recipe fails here instead of shipping that binary.
#ifndef ALIVE_FILTER_MID_NO_SLP
#error "filter_mid.hip must be compiled with -fno-slp-vectorize -DALIVE_FILTER_MID_NO_SLP (see csrc/Makefile)"
#endif

namespace {

constexpr int C = 64;
constexpr int HALO = 56;
constexpr int NCONV = 6;
constexpr int BL = 256;                 // columns per tile incl. halo
constexpr int TT = BL - HALO;           // 200 output columns per tile
constexpr int NCT = BL / 16;            // 16 column tiles
constexpr int NFP = 8;                  // FiLM frames staged per tile
constexpr int NFS = NFP + 1;            // row pitch of the staged FiLM table: the four channel groups of a column (lanes kq = 0..3,
                                        // 4 rows apart) land on different banks (pitch 8 put them 128 B apart: 4-way conflicts)
constexpr int ROWB = 128;               // bytes per LDS row (64 channels bf16)
constexpr int PLANE = BL * ROWB;        // 32 KB
constexpr int BUF = 2 * PLANE;          // hi + lo
constexpr int W_IN = 2 * C * C;         // bf16 elements of the input conv [2][64][64]
constexpr int W_K5 = 2 * C * 5 * C;     // bf16 elements of a k5 conv [2][64][320]
constexpr int LDS_BYTES = 2 * BUF + NCONV * 2 * C * NFS * 4 + BL * 8;

__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, h);
}

__global__ __launch_bounds__(256, 1) void filter_block64_kernel(const float* __restrict__ U, int L,
                                                                const unsigned short* __restrict__ W16,
                                                                const float* __restrict__ biases,
                                                                const float* __restrict__ film, int film_rows, int Lf,
                                                                int film_off, float ratio, int t_off, int f_off, int film_ld, const float* __restrict__ skip,
                                                                float* __restrict__ out, long long* stamps) {
#ifdef ALIVE_STAMPS                 // diagnostic build only (make EXTRA=-DALIVE_STAMPS; tools/bench_filter_mid.py)
#define STAMP(i) ts[i] = wall_clock64()
    long long ts[4], tw = 0;
#else
#define STAMP(i)
#endif
    STAMP(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* bufZ = sm;
    unsigned char* bufY = sm + BUF;
    float* Fs = (float*)(sm + 2 * BUF);               // [NCONV][2][C][NFS]
    uint2* Xc = (uint2*)(Fs + NCONV * 2 * C * NFS);   // [BL] interpolation coordinates of a column: (i0 | i1 << 16, w1)

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, kq = lane >> 4;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * TT;
    const int tbase = t0 - HALO;
    const float* Un = U + (size_t)n * C * L;

    // ---- FiLM rows of the tile ----
    const int f_lo = lerp_coord((tbase < 0 ? 0 : tbase) + t_off, ratio, Lf).i0;     // frames of the WINDOW (t_off: range mode)
    for (int e = tid; e < NCONV * 2 * C * NFP; e += 256) {
        const int f = e % NFP, c = (e / NFP) % C, sel = (e / (NFP * C)) & 1, q = e / (NFP * C * 2);
        int fr = f_lo + f;
        fr = fr < Lf ? fr : Lf - 1;
        int fc = fr - f_off;                             // frame of the window -> column of the film tensor
        fc = fc < 0 ? 0 : (fc < film_ld ? fc : film_ld - 1);
        Fs[((q * 2 + sel) * C + c) * NFS + f] = film[((size_t)n * film_rows + film_off + q * 2 * C + sel * C + c) * film_ld + fc];
    }
    {   // F.interpolate coordinates of this thread's column, relative to the staged FiLM frames
        int t = tbase + tid;
        t = t < 0 ? 0 : (t < L ? t : L - 1);
        const Lerp lp = lerp_coord(t + t_off, ratio, Lf);
        int i0 = lp.i0 - f_lo, i1 = lp.i1 - f_lo;
        i0 = i0 < NFP - 1 ? i0 : NFP - 1;
        i1 = i1 < NFP - 1 ? i1 : NFP - 1;
        Xc[tid] = make_uint2((unsigned)i0 | ((unsigned)i1 << 16), __float_as_uint(lp.w1));
    }
    // ---- stage the raw input tile, split into planes: thread = column; all 64 channel loads are in flight together ----
    {
        const int t = tbase + tid;
        const bool ok = t >= 0 && t < L;
        const float* uc = Un + (ok ? t : 0);
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = uc[(size_t)c * L];
#pragma unroll
        for (int ck = 0; ck < 8; ++ck) {
            u32x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = ok ? v[ck * 8 + 2 * e] : 0.0f, x1 = ok ? v[ck * 8 + 2 * e + 1] : 0.0f;
                const unsigned h = pack2(x0, x1);
                hi[e] = h;
                lo[e] = pack2(x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xffff0000u));
            }
            unsigned char* dst = bufZ + tid * ROWB + ((ck ^ (tid & 7)) << 4);
            *(u32x4*)dst = hi;
            *(u32x4*)(dst + PLANE) = lo;
        }
    }
    __syncthreads();
    STAMP(1);

    // residual stream: h[ct][e] = channel 16 w + 4 kq + e of column 16 ct + c16 (MFMA C layout)
    f32x4 h[NCT];

    // FiLM operands of one column tile for this lane: interpolation weight and, per channel pair, the scale / shift
    // values at the two frames -- fetched one tile AHEAD of their use, so the epilogue has no LDS round trip in it
    struct Film {
        float w1;
        f32x2 s0[2], s1[2], h0[2], h1[2];
    };
    auto film_fetch = [&](int q, int col, Film& F) {
        const uint2 xc = Xc[col];
        const int i0 = xc.x & 0xffff, i1 = xc.x >> 16;
        F.w1 = __uint_as_float(xc.y);
        const float* f = Fs + ((q * 2) * C + 16 * w + 4 * kq) * NFS;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float* fa = f + (2 * e) * NFS;
            const float* fb = fa + NFS;
            F.s0[e] = f32x2{fa[i0], fb[i0]};
            F.s1[e] = f32x2{fa[i1], fb[i1]};
            F.h0[e] = f32x2{fa[C * NFS + i0], fb[C * NFS + i0]};
            F.h1[e] = f32x2{fa[C * NFS + i1], fb[C * NFS + i1]};
        }
    };
    // gelu -> FiLM of the next conv's input for the 4 channels of this lane at column `col`, re-split and stored as
    // planes; two channels per instruction on the packed fp32 pipe
    auto modulate_store = [&](const Film& F, unsigned char* dst, int col, const f32x4& v) {
        const float w1 = F.w1, w0 = 1.0f - w1;
        f32x2 z[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const f32x2 sc = pk_fma(pk_splat(w0), F.s0[e], pk_splat(w1) * F.s1[e]);    // fma(w0, a, round(w1 * b)): ATen's linear interp
            const f32x2 sh = pk_fma(pk_splat(w0), F.h0[e], pk_splat(w1) * F.h1[e]);
            z[e] = gelu_fast2(f32x2{v[2 * e], v[2 * e + 1]}) * sc + sh;
        }
        const unsigned h01 = pack2(z[0][0], z[0][1]), h23 = pack2(z[1][0], z[1][1]);
        const f32x2 r0 = z[0] - f32x2{__uint_as_float(h01 << 16), __uint_as_float(h01 & 0xffff0000u)};
        const f32x2 r1 = z[1] - f32x2{__uint_as_float(h23 << 16), __uint_as_float(h23 & 0xffff0000u)};
        const unsigned l01 = pack2(r0[0], r0[1]), l23 = pack2(r1[0], r1[1]);
        // channels 16 w + 4 kq .. +3 -> chunk 2 w + (kq >> 1), byte 8 (kq & 1) inside it
        unsigned char* p = dst + col * ROWB + (((2 * w + (kq >> 1)) ^ (col & 7)) << 4) + 8 * (kq & 1);
        *(uint2*)p = make_uint2(h01, h23);
        *(uint2*)(p + PLANE) = make_uint2(l01, l23);
    };

    // one conv over the whole tile: KS k-steps of 32 channels, A fragments (this wave's 16 rows) in registers
    // B fragment of (row r, ci-block cb): 16 B at r * 128 + ((4 cb + kq) ^ (r & 7)) * 16, planes PLANE apart
    // ---- input_conv (1x1, K = 64 = 2 k-steps): h = Win * U + b ; z0 = mod_0(h) written over the same columns ----
    {
        bf16x8 a[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                a[s][pl] = *(const bf16x8*)(W16 + (size_t)pl * C * C + (size_t)(16 * w + c16) * C + s * 32 + 8 * kq);
        f32x4 b4;
#pragma unroll
        for (int e = 0; e < 4; ++e) b4[e] = biases[16 * w + 4 * kq + e];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int r = ct * 16 + c16;
            f32x4 acc = b4;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned char* bp = bufZ + r * ROWB + (((4 * s + kq) ^ (r & 7)) << 4);
                const bf16x8 bh = *(const bf16x8*)bp, bl = *(const bf16x8*)(bp + PLANE);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][1], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][0], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][0], bh, acc, 0, 0, 0);
            }
            h[ct] = acc;
        }
    }
    __syncthreads();                       // every wave has read the raw columns
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        Film F;
        film_fetch(0, ct * 16 + c16, F);
        modulate_store(F, bufZ, ct * 16 + c16, h[ct]);
    }
    __syncthreads();

    STAMP(2);
    // ---- three FilterResBlocks: q = 2 j (c1), 2 j + 1 (c2), dilation 2^j    (decoder.py:128-134) ----
    const bool first_tile = tbase < 0;                       // block-uniform: ReflectionPad1d applies (common.py:88)
#pragma unroll 1
    for (int q = 0; q < NCONV; ++q) {
#ifdef ALIVE_STAMPS
        const long long tq0 = wall_clock64();
#endif
        const unsigned short* Wq = W16 + W_IN + (size_t)q * W_K5;
        bf16x8 a[10][2];                   // k-step s = 2 j + cb (tap j, channel block cb): k = j * 64 + cb * 32 + 8 kq
#pragma unroll
        for (int s = 0; s < 10; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                a[s][pl] = *(const bf16x8*)(Wq + (size_t)pl * C * 5 * C + (size_t)(16 * w + c16) * (5 * C) + s * 32 + 8 * kq);
        f32x4 b4;
#pragma unroll
        for (int e = 0; e < 4; ++e) b4[e] = biases[(1 + q) * C + 16 * w + 4 * kq + e];
        const int d = 1 << (q >> 1);
        const bool second = q & 1;
        const unsigned char* in = second ? bufY : bufZ;
        unsigned char* dst = second ? bufZ : bufY;

        // B fragment of k-step s = 2 j + cb of column tile ct: row r = 16 ct + c16 + (j - 4) d, 16 B at
        // r * 128 + ((4 cb + kq) ^ (r & 7)) * 16.  (r & 7) does not depend on ct, so the address is
        // base[cb][j] + 2048 ct: one register per (tap, channel block) and an immediate per column tile.
        int base[2][5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int r0 = c16 + (j - 4) * d;
            base[0][j] = r0 * ROWB + ((kq ^ (r0 & 7)) << 4);
            base[1][j] = base[0][j] ^ 64;
        }
        bf16x8 fh[10], fl[10];
        auto load_frags = [&](int ct) {
            // column tile 0 reaches left of the tile (rows that are never valid: clamp), and the first tile of a window
            // reflects at t = 0 for the columns below 16 d + HALO: those take the per-lane form
            if (ct == 0 || (first_tile && ct < 5)) {
                const int col = ct * 16 + c16;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    int ta = tbase + col + (j - 4) * d;
                    ta = ta < 0 ? -ta : ta;
                    int r = ta - tbase;
                    r = r < 0 ? 0 : (r < BL ? r : BL - 1);
                    const int o = r * ROWB + ((kq ^ (r & 7)) << 4);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const unsigned char* bp = in + (o ^ (cb << 6));
                        fh[2 * j + cb] = *(const bf16x8*)bp;
                        fl[2 * j + cb] = *(const bf16x8*)(bp + PLANE);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const unsigned char* bp = in + base[cb][j] + ct * (16 * ROWB);
                        fh[2 * j + cb] = *(const bf16x8*)bp;
                        fl[2 * j + cb] = *(const bf16x8*)(bp + PLANE);
                    }
            }
        };
        load_frags(0);
#ifdef ALIVE_STAMPS
        __builtin_amdgcn_s_waitcnt(0);
        tw += wall_clock64() - tq0;
#endif
        // The column loop is unrolled (residual registers and LDS immediates are static) and software-pipelined: the
        // MFMAs of tile ct + 1 are issued BEFORE the epilogue of tile ct, so the epilogue's VALU / LDS work sits in the
        // shadow of a dependent MFMA chain instead of behind it (one wave per SIMD: nothing else would fill it).
        // One fragment buffer: the fragments of tile ct + 2 are requested right behind the MFMAs of tile ct + 1.
        auto mma_tile = [&](f32x4& acc0, f32x4& acc1) {
            acc0 = b4;
            acc1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#ifdef ALIVE_FB64_3CHAINS              // DIAGNOSTIC build only (tools/repro_filter_block64.sh): the two cross terms on accumulators of
                                       // their own.  Slower (7.58 against 7.40 ms) and -- with the per-tile fence and -fno-slp-vectorize
                                       // in place -- NOT deterministic: 19 of 20 launches differ from the first (DESIGN.md 3.2b').
            f32x4 acc2 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][1], fh[s], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][0], fl[s], acc2, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][0], fh[s], acc0, 0, 0, 0);
            }
            acc1 = acc1 + acc2;
#else
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][1], fh[s], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][0], fl[s], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][0], fh[s], acc0, 0, 0, 0);
            }
#endif
        };
        // the last conv has no consumer for the modulated output: it runs the same code with the FiLM rows of conv 5 and
        // writes a tile nobody reads, which keeps the column body free of branches (one basic block per tile)
        const int qn = q + 1 < NCONV ? q + 1 : NCONV - 1;
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 p0, p1, n0, n1;
        Film Fc, Fn;
        film_fetch(qn, c16, Fc);
        mma_tile(p0, p1);
        load_frags(1);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            if (ct + 1 < NCT) {
                film_fetch(qn, (ct + 1) * 16 + c16, Fn);
                mma_tile(n0, n1);
                if (ct + 2 < NCT) load_frags(ct + 2);
            }
            f32x4 v = (p0 + p1) + (second ? h[ct] : zero4);
            h[ct] = second ? v : h[ct];
            modulate_store(Fc, dst, ct * 16 + c16, v);
            // The fence is load-bearing: with the 16 column bodies merged into one scheduling region hipcc (ROCm 7.2) produced a
            // schedule whose results differed from run to run (1.5e-2 off); with one region per column tile they are exact.
#if !defined(ALIVE_NO_TILE_FENCE)    // -DALIVE_NO_TILE_FENCE[=n]: diagnostic builds only (tools/diag_filter_nofence.py, DESIGN.md 3.2b')
            __builtin_amdgcn_sched_barrier(0);
#elif ALIVE_NO_TILE_FENCE == 1       // compiler-level memory clobber instead of the scheduling fence
            asm volatile("" ::: "memory");
#elif ALIVE_NO_TILE_FENCE == 2       // fence only behind the last four column tiles
            if (ct >= 12) __builtin_amdgcn_sched_barrier(0);
#elif ALIVE_NO_TILE_FENCE == 3       // fence only behind the first twelve
            if (ct < 12) __builtin_amdgcn_sched_barrier(0);
#endif
            p0 = n0;
            p1 = n1;
            Fc = Fn;
        }
        __syncthreads();
    }

    STAMP(3);
    // ---- store the tile (+ U-Net skip, decoder.py:191): through LDS so that global accesses are 16-B vectors along t ----
    float* Ht = (float*)sm;                           // [64][BL + 4] fp32 = 66.5 KB over bufZ / bufY
    constexpr int HP = BL + 4;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) Ht[(16 * w + 4 * kq + e) * HP + ct * 16 + c16] = h[ct][e];
    __syncthreads();
    constexpr int NV = (C * (TT / 4) + 255) / 256;          // 13 vectors of 4 columns per thread
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
        const f32x4 v = *(const f32x4*)&Ht[co * HP + HALO + c4];
        if (t + 3 < L) {
            *(f32x4*)(out + o) = v + sk[i];
        } else {
            for (int e = 0; e < 4 && t + e < L; ++e) out[o + e] = v[e] + (skip != nullptr ? skip[o + e] : 0.0f);
        }
    }
#ifdef ALIVE_STAMPS
    if (stamps != nullptr && tid == 0) {
        long long* o = stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        o[0] = ts[1] - ts[0]; o[1] = ts[2] - ts[1]; o[2] = ts[3] - ts[2]; o[3] = wall_clock64() - ts[3]; o[4] = tw;
    }
#endif
}

}  // namespace

static long long* g_stamps64 = nullptr;
extern "C" void alive_debug_set_stamps64(long long* p) { g_stamps64 = p; }
extern "C" int64_t alive_filter_block64_weights(void) { return (int64_t)W_IN + (int64_t)NCONV * W_K5; }

extern "C" int alive_filter_block64(const float* U, int N, int L, const void* W16, const float* biases, const float* film,
                                    int film_rows, int Lf, int film_off, const float* skip, float* out, void* stream) {
    return alive_filter_block64_range(U, N, L, W16, biases, film, film_rows, Lf, film_off, 0, 0, Lf, skip, out, stream);
}

extern "C" int alive_filter_block64_range(const float* U, int N, int L, const void* W16, const float* biases, const float* film,
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
        hipError_t e = optin.ensure({(const void*)filter_block64_kernel}, LDS_BYTES);
        if (e != hipSuccess) {
            alive_set_error("alive_filter_block64: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(e));
            return ALIVE_ERR_LAUNCH;
        }
    }
    const float ratio = (float)film_ld / (float)L;       // == window frames / window samples at this rate
    dim3 g(cdiv(L, TT), N);
    filter_block64_kernel<<<g, 256, LDS_BYTES, (hipStream_t)stream>>>(U, L, (const unsigned short*)W16, biases, film, film_rows,
                                                                     Lf, film_off, ratio, t0, f0, film_ld, skip, out, g_stamps64);
    ALIVE_CHECK_LAUNCH("alive_filter_block64");
    return ALIVE_OK;
}
