// Cosine kNN regression against the voice library -- replaces match_features
// (/root/reference/module/common.py:96-109) and VoiceLibrary.match (voice_library.py:15-33).
//
// The reference materialises cos[T][M] with an fp32 bmm and runs topk over it.
// Here the [T][M] matrix never exists.  A search is a chain of tiers, launched up front and gated by device-side counters:
//   probe  (big batches) the fp8 stage + its certificate on 1 024 sample frames decides whether the batch starts on fp8 or bf16.
//   0. knn_score8_kernel  candidate stage on the block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4): frames
//      stationary in registers, library tiles streamed through LDS by LDS-DMA, lane-private top-16 lists per half-wave;
//      a tile's accumulators are copied out and folded under the NEXT tile's MFMAs (two AGPR sets, explicit late reads);
//      in big batches the blocks of a library split start their lists at the seeds the previous split left (SeedArgs).
//   1. knn_score_kernel   the same structure on the bf16 MFMA (v_mfma_f32_32x32x16_bf16), lists of 8: first stage of
//      alive_knn_search, and the re-search of the frames whose fp8 candidate set could not be certified (up to 64 such
//      frames go straight to 3.); <true>: the collect form (fixed per-frame threshold) behind a failed bf16 certificate.
//   2. knn_rescore_kernel every surviving candidate is re-scored in fp32 with the reference's arithmetic
//      (normalise-then-dot), the exact top-k is taken, and the frame is CERTIFIED against the candidate stage's score
//      error measured on its own candidates (the MFMA passes only have to be superset generators).
//   3. knn_exact_kernel   frames neither certificate passes, and every search with k > 8: brute-force fp32 scan with the
//      rescoring arithmetic.  knn_scan_kernel is its streaming form (a handful of frames, no candidate stage at all).
//   4. knn_merge_gather_kernel merges per-shard exact lists (after an RCCL all-gather when the library is sharded),
//      gathers the k rows, mean, blend.
//
// Layout: frames on the MFMA column/lane axis (B operand), library rows on the
// row/register axis (A operand): a lane owns two frame columns and sees 32
// library rows of each per tile, so list maintenance is lane-private.
#include "common.h"

namespace {

constexpr int D = ALIVE_DIM;          // 768
constexpr int KP = ALIVE_KPRIME;      // 16
constexpr int TILE = 128;             // library row padding granule
constexpr int MAX_SPLIT = 64;         // max library splits (grid.y); candidates/frame = split*KP <= 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// x / n for the operands the exact tiers divide -- the rows and norms of a packed library: finite, normal range, n > 0 (zero-norm rows
// are refused when a library is packed).  This IS the fp32 division hipcc expands `x / n` into (v_rcp_f32, one Newton step on the
// reciprocal, the quotient with two residual corrections) WITHOUT its v_div_scale / v_div_fmas / v_div_fixup steps, which only act on
// operands near the ends of the exponent range: bit for bit the IEEE quotient for every operand pair these kernels see, at 5 fused
// operations per element and 3 per divisor instead of 11 + a quarter-rate reciprocal per element (round 6: the rescoring kernel spent
// 2 400 of its ~7 000 vector instructions per frame in divisions).  ONE helper for knn_rescore_kernel, knn_exact_kernel and
// knn_scan_kernel: their values must agree bitwise with each other.
struct NormDiv {
    float n, y;
};
__device__ __forceinline__ NormDiv norm_div(float n) {
    float y = __builtin_amdgcn_rcpf(n);
    const float e = fmaf(-n, y, 1.0f);
    y = fmaf(e, y, y);
    return NormDiv{n, y};
}
__device__ __forceinline__ float div_by(float x, const NormDiv& d) {
    float q = x * d.y;
    float r = fmaf(-d.n, q, x);
    q = fmaf(r, d.y, q);
    r = fmaf(-d.n, q, x);
    return fmaf(r, d.y, q);
}

// DPP forms of the order-independent wave reductions (max / min): no LDS permutes (a __shfl_xor is a ds_bpermute_b32: the rescoring
// kernel issued ~650 of them per frame).  row = 16 consecutive lanes.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
// min over aligned groups of 8 (LEN8) or 16 lanes, in every lane of the group
template <bool LEN16>
__device__ __forceinline__ float group_min(float v) {
    v = fminf(v, dpp_f<DPP_XOR1>(v));
    v = fminf(v, dpp_f<DPP_XOR2>(v));
    v = fminf(v, dpp_f<DPP_HALF_MIRROR>(v));
    if (LEN16) v = fminf(v, dpp_f<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float wave_max_u(float v) {          // wave-uniform maximum
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ unsigned wave_min_u(unsigned v) {    // wave-uniform minimum of unsigned values
    unsigned o;
    o = dpp_u<DPP_XOR1>(v); v = o < v ? o : v;
    o = dpp_u<DPP_XOR2>(v); v = o < v ? o : v;
    o = dpp_u<DPP_HALF_MIRROR>(v); v = o < v ? o : v;
    o = dpp_u<DPP_MIRROR>(v); v = o < v ? o : v;
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
    return a < b ? a : b;
}

// ----------------------------------------------------------------------------------------------
// packing
// ----------------------------------------------------------------------------------------------
// tokens[D][M] -> norms[M], rows_f32[M][D], lib_bf16[M_pad][D] (normalised).  64 columns per block.
__global__ __launch_bounds__(256) void lib_pack_kernel(const float* __restrict__ tok, int64_t M, int64_t M_pad,
                                                       unsigned short* __restrict__ lib, float* __restrict__ rows,
                                                       float* __restrict__ norms) {
    __shared__ float red[4][64];
    __shared__ float tile[64][65];
    __shared__ float nrm[64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int64_t m = m0 + lane;
    const bool ok = m < M;
    float ss = 0.0f;
    for (int d = wv; d < D; d += 4) {
        float v = ok ? tok[(size_t)d * M + m] : 0.0f;
        ss = fmaf(v, v, ss);
    }
    red[wv][lane] = ss;
    __syncthreads();
    if (wv == 0) {
        float nn = sqrtf(red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
        nrm[lane] = nn;
        if (ok) norms[m] = nn;
    }
    __syncthreads();
    for (int d0 = 0; d0 < D; d0 += 64) {
        for (int r = wv; r < 64; r += 4) tile[r][lane] = ok ? tok[(size_t)(d0 + r) * M + m] : 0.0f;
        __syncthreads();
        for (int r = wv; r < 64; r += 4) {      // r = library row inside the block, lane = feature
            const int64_t mm = m0 + r;
            if (mm < M_pad) {
                float v = tile[lane][r];
                if (mm < M) rows[(size_t)mm * D + d0 + lane] = v;
                float q = (mm < M) ? v / nrm[r] : 0.0f;
                lib[(size_t)mm * D + d0 + lane] = f32_to_bf16_rn(q);
            }
        }
        __syncthreads();
    }
}

// Strict (deterministic) certificate, library side: max over the rows of || r^ - bf16(r^) ||_2, r^ = row / norm as the
// packing kernel computes it.  One wave per row; the maximum lands in bound[0] through an atomicMax on the float's bits
// (non-negative floats order like unsigned integers).  The fp32 sum of squares is rounded up by 1e-4 relative.
__global__ __launch_bounds__(256) void lib_rounding_bound_kernel(const unsigned short* __restrict__ lib, const float* __restrict__ rows,
                                                                 const float* __restrict__ norms, int64_t M,
                                                                 unsigned* __restrict__ bound) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float nn = norms[m];
    float ss = 0.0f;
    for (int d = lane; d < D; d += 64) {
        const float q = rows[(size_t)m * D + d] / nn;
        const float b = __uint_as_float((unsigned)lib[(size_t)m * D + d] << 16);
        const float e = q - b;
        ss = fmaf(e, e, ss);
    }
    ss = wave_sum(ss);
    if (lane == 0) atomicMax(bound, __float_as_uint(sqrtf(ss) * 1.0001f));
}

// ---- fp8 (OCP e4m3) form of the scoring operands: value * 2^8, so that the elements of unit vectors (|x| ~ 0.04)
// sit in e4m3's normal range (2^-6 .. 448); scores come out scaled by 2^16, which a ranking does not see
constexpr float F8_SCALE = 256.0f;
// ---- fp6 (OCP e2m3) form, round 5: value * 2^5 -- sigma of an element ~ 1.15 on a grid of 1/8 up to 2, 1/4 up to 4, 1/2 up to the
// largest value 7.5.  For unit vectors of Gaussian-like elements that grid is as good as e4m3's three mantissa bits (what e4m3
// adds is dynamic range, which normalised rows do not need): the stage's score error is 1.8e-3 in cosine against 1.35e-3
// (simulated on CPU, measured by the certificate on the device), and the MFMA runs at twice the fp8 rate.  Same tile image: every
// group of 32 features keeps its 32-byte slot, the 32 six-bit codes fill its first 24 bytes as a little-endian bit stream.
constexpr float F6_SCALE = 32.0f;
// OCP fp6 e2m3 code of y (round to nearest even, saturating at 7.5): sign | exponent (2) | mantissa (3), bias 1
__device__ __forceinline__ unsigned fp6_e2m3(float y) {
    const unsigned sgn = (__float_as_uint(y) >> 31) << 5;
    const float a = fminf(fabsf(y), 7.5f);
    unsigned code;
    if (a < 1.0f) {
        code = (unsigned)__builtin_rintf(a * 8.0f);                    // subnormals m / 8; 8 = the code of 1.0
    } else {
        unsigned b = __float_as_uint(a);
        b += 0x7FFFFu + ((b >> 20) & 1u);
        code = (b >> 20) - (126u << 3);
        code = code < 31u ? code : 31u;
    }
    return sgn | code;
}
// bf16[n32][32] -> [n32][32 bytes]: 32 codes in the first 24 bytes, 8 bytes of zeros.  Groups at or beyond n_valid (frame rows
// past the bf16 image's padded end) are written as zeros without reading.
// clip (optional, [rows]): set to 1 for every row (24 groups) with an element beyond e2m3's largest value.
__global__ __launch_bounds__(256) void to_fp6_kernel(const unsigned short* __restrict__ in, int64_t n32, int64_t n_valid, u32x4* __restrict__ out,
                                                     unsigned char* __restrict__ clip = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n32) return;
    if (i >= n_valid) {
        out[2 * i] = u32x4{0u, 0u, 0u, 0u};
        out[2 * i + 1] = u32x4{0u, 0u, 0u, 0u};
        return;
    }
    unsigned long long acc = 0;
    int nb = 0, wi = 0;
    unsigned wd[6];
    bool clipped = false;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float v = __uint_as_float((unsigned)in[i * 32 + j] << 16) * F6_SCALE;
        clipped = clipped || fabsf(v) > 7.75f;                         // (7.5 .. 7.75 rounds to 7.5: not a clip)
        acc |= (unsigned long long)fp6_e2m3(v) << nb;
        nb += 6;
        if (nb >= 32) { wd[wi++] = (unsigned)acc; acc >>= 32; nb -= 32; }
    }
    out[2 * i] = u32x4{wd[0], wd[1], wd[2], wd[3]};
    out[2 * i + 1] = u32x4{wd[4], wd[5], 0u, 0u};
    if (clip != nullptr && clipped) clip[i / (D / 32)] = 1;
}
__device__ __forceinline__ unsigned char to_fp8(float v) {
    return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v * F8_SCALE, 0.0f, 0, false) & 0xff);
}
__device__ __forceinline__ unsigned to_fp8x4(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a * F8_SCALE, b * F8_SCALE, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c * F8_SCALE, d * F8_SCALE, w, true);
    return (unsigned)w;
}

// lib_bf16[M_pad][D] (normalised rows) -> lib_f8[M_pad][D]; 8 elements per thread
__global__ __launch_bounds__(256) void lib_to_fp8_kernel(const unsigned short* __restrict__ lib, int64_t n8, uint2* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const u32x4 v = ((const u32x4*)lib)[i];
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = __uint_as_float(v[j] << 16);
        f[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
    }
    out[i] = make_uint2(to_fp8x4(f[0], f[1], f[2], f[3]), to_fp8x4(f[4], f[5], f[6], f[7]));
}

// s_bf16[Tt_pad][D] -> s_f8[Tt_pad][D]
__global__ __launch_bounds__(256) void src_to_fp8_kernel(const unsigned short* __restrict__ s_bf16, int64_t n8, uint2* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const u32x4 v = ((const u32x4*)s_bf16)[i];
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = __uint_as_float(v[j] << 16);
        f[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
    }
    out[i] = make_uint2(to_fp8x4(f[0], f[1], f[2], f[3]), to_fp8x4(f[4], f[5], f[6], f[7]));
}

// ---- the ROTATED form of the fp8 stage operands (round 6) -------------------------------------------------------------
// A dense bank (one speaker: every row shares a large common component) defeats an 8-bit candidate stage: thousands of rows sit inside
// the stage's score error of a frame's k-th neighbour, and 172 800 of 172 800 bench frames fell through to the bf16 stage.  Measured
// before anything was built (tools/knn_pca_probe.py, profiles/r06_knn_pca_probe.json): with the bank's leading principal directions
// carried at 8 significant bits instead of 4 the stage error falls from 1.6e-3 to 3.7e-4 and 93 % of the frames certify.  No kernel
// change is needed for that on e4m3, whose range makes every code `value x 2^8` as before:
//   * the bank lives in a subspace (the reference's ContentEncoder ends in Conv1d(512 -> 768): rank <= 513); unit vectors are expressed
//     in a basis W = [U_0..63 | U_64..575 R] -- the 64 leading eigenvectors of the rows' second moment, then the next 512 mixed by a
//     fixed random rotation R (eigen-coordinates concentrate the residual's energy in a few blocks, which averages the rounding errors
//     over fewer terms: sigma 5.9e-4 unmixed against 3.7e-4) -- the 192 directions behind them carry no energy and are dropped;
//   * the 64 alpha coordinates travel as TWO e4m3 digits each (hi = e4m3(a), lo = e4m3(a - hi)), laid out so that the plain dot product
//     of the two code vectors IS (hi + lo)(hi + lo) + rho . rho:   rows   [hi | lo | hi | lo | rho x 507 | 5 codes]
//                                                                  frames [hi | hi | lo | lo | rho x 507 | 5 codes]      (64 codes per group)
//   * the LEADING coordinate alpha_0 is ~0.65 for every row and every frame (it IS the common component), and two e4m3 digits of 0.65 are
//     good to ~2^-9 of it: that one coordinate carried two thirds of the stage's remaining error (sigma 3.7e-4 -> 2.4e-4 with it exact,
//     certified frames 92.9 % -> 99.6 %: tools/knn_pca_probe.py "A0").  It is CENTRED: alpha_0 = c0 + a with c0 an e4m3-exact constant of
//     the bank (the mean of its rows' alpha_0), a in two digits in alpha_0's four slots, and
//     (c0 + a_q)(c0 + a_r) = a_q a_r + [c0 c0 + a_r c0 + c0 a_q] through the vector's last five codes:
//                                                                  rows   [c0, a_hi, a_lo, c0, c0]      frames [c0, c0, c0, a_hi, a_lo]
// y: the rotated coordinates (576 per vector; the last 5 are not read), `side` 0 = library rows, row-major [n][RC]; side 1 = frames, as alive_conv1d leaves them: [N][RC][T]
// (frame f = n T + t), NOT normalised -- the kernel divides by the frame's norm (|W^T x| = |x| for a frame inside the bank's subspace;
// a frame with energy outside it gets scores that are too high by one factor for all rows: the ranking stands, the certificate sees
// the mismatch with the exact cosines and sends the frame to the next tier).  out: [n_pad][768] codes, zero rows beyond n.
constexpr int ROT_A = 64, ROT_RHO = 507, ROT_C = 576;      // 64 + 507 coordinates read (of 576 passed) -> 4 x 64 + 507 + 5 = 768 codes
__device__ __forceinline__ float fp8_value(unsigned code) {             // e4m3 byte -> the value it stands for (x 2^-8 applied by the caller)
    return __builtin_amdgcn_cvt_f32_fp8((int)code, 0);
}
__global__ __launch_bounds__(256) void rot_codes_kernel(const float* __restrict__ y, int64_t n, int64_t n_pad, int T, int side, float c0,
                                                        unsigned char* __restrict__ out) {
    __shared__ float tile[ROT_C][33];          // [coordinate][frame of the block], 76 KB
    __shared__ float ssq[8][32];
    const int fl = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int64_t f = (int64_t)blockIdx.x * 32 + fl;
    const bool live = f < n;
    const float* base;
    int64_t sc;
    if (T > 0) {                               // frames: [N][RC][T]
        const int64_t ni = live ? f / T : 0, ti = live ? f - ni * T : 0;
        base = y + (size_t)ni * ROT_C * T + ti;
        sc = T;
    } else {                                   // rows: [n][RC]
        base = y + (size_t)(live ? f : 0) * ROT_C;
        sc = 1;
    }
    float s2 = 0.0f;
    for (int c = cg; c < ROT_C; c += 8) {
        const float v = live ? base[(size_t)c * sc] : 0.0f;
        tile[c][fl] = v;
        s2 = fmaf(v, v, s2);
    }
    ssq[cg][fl] = s2;
    __syncthreads();
    float nrm = 0.0f;
#pragma unroll
    for (int g = 0; g < 8; ++g) nrm += ssq[g][fl];
    const float inv = (T > 0 && nrm > 0.0f) ? 1.0f / sqrtf(nrm) : (T > 0 ? 0.0f : 1.0f);     // rows arrive as coordinates of unit vectors
    if (f >= n_pad) return;
    // 24 groups of 32 codes per vector: groups 0 .. 7 are the four alpha blocks (64 codes each), 8 .. 23 rho
    for (int g = cg; g < 24; g += 8) {
        unsigned w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * q + e;                       // code j of group g
                float x;
                auto lo_digit = [](float al) {                                    // al - e4m3(al): what the second digit carries
                    const unsigned ch = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(al * F8_SCALE, 0.0f, 0, false) & 0xffu;
                    return al - fp8_value(ch) * (1.0f / F8_SCALE);
                };
                if (g < 8) {
                    const int blk = g >> 1, a = 32 * (g & 1) + j;                  // alpha coordinate a of digit block blk
                    const float al = tile[a][fl] * inv - (a == 0 ? c0 : 0.0f);     // (the leading coordinate is centred)
                    const bool lo = side == 0 ? (blk & 1) != 0 : blk >= 2;         // rows [hi lo hi lo], frames [hi hi lo lo]
                    x = lo ? lo_digit(al) : al;
                } else {
                    const int r = 32 * (g - 8) + j;                                // rho position 0 .. 511
                    if (r < ROT_RHO) {
                        x = tile[ROT_A + r][fl] * inv;
                    } else {                                                       // the five codes of the centred leading coordinate
                        const int q = r - ROT_RHO;
                        const float a0 = tile[0][fl] * inv - c0;
                        const int role = side == 0 ? (q == 1 ? 1 : (q == 2 ? 2 : 0)) : (q == 3 ? 1 : (q == 4 ? 2 : 0));       // 0: c0, 1: a_hi, 2: a_lo
                        x = role == 0 ? c0 : (role == 1 ? a0 : lo_digit(a0));
                    }
                }
                v[e] = live ? x : 0.0f;
            }
            w[q] = to_fp8x4(v[0], v[1], v[2], v[3]);
        }
        u32x4* o = (u32x4*)(out + (size_t)f * D + 32 * g);
        o[0] = u32x4{w[0], w[1], w[2], w[3]};
        o[1] = u32x4{w[4], w[5], w[6], w[7]};
    }
}

// src[N][D][T] -> s_f32[Tt][D] (normalised, fp32), s_bf16[Tt_pad][D]
// dq (optional): || q^ - bf16(q^) ||_2 per frame, rounded up -- the frame's share of the strict certificate's bound
__global__ __launch_bounds__(256) void src_prep_kernel(const float* __restrict__ src, int T, int64_t Tt, int64_t Tt_pad,
                                                       float* __restrict__ s_f32, unsigned short* __restrict__ s_bf16,
                                                       float* __restrict__ dq) {
    __shared__ float red[4][64];
    __shared__ float tile[64][65];
    __shared__ float nrm[64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t f0 = (int64_t)blockIdx.x * 64;
    const int64_t ft = f0 + lane;
    const bool ok = ft < Tt;
    const int64_t n = ok ? ft / T : 0;
    const int t = ok ? (int)(ft - n * T) : 0;
    const float* col = src + (size_t)n * D * T + t;
    float ss = 0.0f;
    for (int d = wv; d < D; d += 4) {
        float v = ok ? col[(size_t)d * T] : 0.0f;
        ss = fmaf(v, v, ss);
    }
    red[wv][lane] = ss;
    __syncthreads();
    if (wv == 0) nrm[lane] = sqrtf(red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
    __syncthreads();
    float e2[16];                                     // rounding error of frame f0 + wv + 4 i, this lane's features
#pragma unroll
    for (int i = 0; i < 16; ++i) e2[i] = 0.0f;
    for (int d0 = 0; d0 < D; d0 += 64) {
        for (int r = wv; r < 64; r += 4) tile[r][lane] = ok ? col[(size_t)(d0 + r) * T] : 0.0f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = wv + 4 * i;
            const int64_t ff = f0 + r;
            if (ff < Tt_pad) {
                // a zero-norm frame (the reference divides 0 / 0 -> NaN cosines, whose top-k is unspecified: common.py:102-105) has
                // cosine 0 against every row here: the search returns val = 0 and, by the tie rule, the k lowest row indices
                float q = (ff < Tt && nrm[r] > 0.0f) ? tile[lane][r] / nrm[r] : 0.0f;
                if (ff < Tt) s_f32[(size_t)ff * D + d0 + lane] = q;
                const unsigned short b = f32_to_bf16_rn(q);
                s_bf16[(size_t)ff * D + d0 + lane] = b;
                const float e = q - __uint_as_float((unsigned)b << 16);
                e2[i] = fmaf(e, e, e2[i]);
            }
        }
        __syncthreads();
    }
    if (dq != nullptr) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float t = wave_sum(e2[i]);
            const int64_t ff = f0 + wv + 4 * i;
            if (lane == 0 && ff < Tt) dq[ff] = sqrtf(t) * 1.0001f;
        }
    }
}

// the same for a handful of frames (streaming): one block per frame, threads along d -- the kernel above walks d
// serially with one frame per lane, which for 8 frames is 2 x 192 dependent strided loads per wave
__global__ __launch_bounds__(256) void src_prep_small_kernel(const float* __restrict__ src, int T, int64_t Tt,
                                                             float* __restrict__ s_f32, unsigned short* __restrict__ s_bf16,
                                                             float* __restrict__ dq) {
    __shared__ float red[4];
    __shared__ float ered[4];
    const int64_t ft = blockIdx.x;
    const int tid = threadIdx.x;
    if (ft >= Tt) {                                  // padding frames of the last 256-frame block: zero rows
        for (int d = tid; d < D; d += 256) s_bf16[(size_t)ft * D + d] = 0;
        return;
    }
    const int64_t n = ft / T;
    const int t = (int)(ft - n * T);
    const float* col = src + (size_t)n * D * T + t;
    float v[D / 256];
    // the norm sums d in the order of the kernel above: four interleaved partial sums (d mod 4), then their sum
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < D / 256; ++i) v[i] = col[(size_t)(tid + 256 * i) * T];
    // partial sum of residue class (tid & 3), accumulated in ascending d like the serial loop: gather through LDS
    __shared__ float xs[D];
#pragma unroll
    for (int i = 0; i < D / 256; ++i) xs[tid + 256 * i] = v[i];
    __syncthreads();
    if (tid < 4) {
        for (int d = tid; d < D; d += 4) ss = fmaf(xs[d], xs[d], ss);
        red[tid] = ss;
    }
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    float e2 = 0.0f;
#pragma unroll
    for (int i = 0; i < D / 256; ++i) {
        const float q = nrm > 0.0f ? v[i] / nrm : 0.0f;          // zero-norm frame: see src_prep_kernel
        s_f32[(size_t)ft * D + tid + 256 * i] = q;
        const unsigned short b = f32_to_bf16_rn(q);
        s_bf16[(size_t)ft * D + tid + 256 * i] = b;
        const float e = q - __uint_as_float((unsigned)b << 16);
        e2 = fmaf(e, e, e2);
    }
    if (dq != nullptr) {                             // block-uniform
        e2 = wave_sum(e2);
        if ((tid & 63) == 0) ered[tid >> 6] = e2;
        __syncthreads();
        if (tid == 0) dq[ft] = sqrtf(ered[0] + ered[1] + ered[2] + ered[3]) * 1.0001f;
    }
}

// ----------------------------------------------------------------------------------------------
// scoring: frames stationary in registers, library streamed through LDS by LDS-DMA
// ----------------------------------------------------------------------------------------------
// One block = 4 waves = 256 source frames (64 per wave).  A wave keeps the bf16 B-operand fragments of
// its 64 frames for ALL of K = 768 in registers (2 x 48 x 4 = 384 VGPRs; one wave per SIMD, 512-register
// budget), so the only operand that moves in the main loop is the library: tiles of 32 rows x 768 k
// (48 KB) are copied global -> LDS by `global_load_lds_dwordx4` (no VGPR round trip, no ds_write) into
// a double buffer while the previous tile is consumed; one barrier per tile (96 MFMAs per wave).
// DMA pieces are 8 rows x 128 B (full lines from L2); the 16-B chunk order inside a line is XOR-swizzled
// on the SOURCE address (the LDS image of an LDS-DMA is lane-linear) and pieces sit at a 1152-B stride,
// which together make every ds_read_b128 of an A fragment conflict-free.
constexpr int KH = KP / 2;                // entries per half-list: each lane keeps its own list per frame column (the two
                                          // half-waves of a column see disjoint library rows), so updates are lane-private
constexpr int FT = 256;                  // frames per block
constexpr int LT = 32;                   // library rows per tile
constexpr int NK16 = D / 16;             // 48 MFMA k-steps
constexpr int PIECE = 1152;              // LDS bytes reserved per 1-KB DMA piece (stagger of 128 B per odd piece)
constexpr int NPIECE = 4 * (D / 64);     // 4 row groups x 12 k-segments = 48
constexpr int ABUF = NPIECE * PIECE;     // 55296 B per tile buffer
constexpr int SCORE_LDS = 2 * ABUF + FT * KP * 8;   // 143360 B

// Fallback kernels of the fp8 search are launched unconditionally (no host sync) and decide on the device whether they
// have work: active iff lo < *cnt <= hi (cnt = number of frames whose fp8 candidate set could not be certified).
constexpr int GATE_BOOL = -2;             // knn_rescore_kernel: gate_lo value that makes gate_cnt an on / off word
__device__ __forceinline__ bool gate_open(const int* cnt, int lo, int hi, int& c) {
    c = 0x7fffffff;
    if (cnt == nullptr) return true;
    c = *cnt;
    return c > lo && c <= hi;
}

// compacts the bf16 operand rows of the flagged frames: out[slot] = s_bf16[list[slot]], zero rows up to the next 256
__global__ __launch_bounds__(128) void gather_frames_kernel(const unsigned short* __restrict__ s_bf16, const int* __restrict__ list,
                                                            const int* __restrict__ cnt, int lo, int hi, unsigned short* __restrict__ out) {
    int c;
    if (!gate_open(cnt, lo, hi, c)) return;
    const int slot = blockIdx.x;
    if (slot >= (c + 255) / 256 * 256) return;
    if (threadIdx.x >= 96) return;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (slot < c) v = ((const u32x4*)(s_bf16 + (size_t)list[slot] * D))[threadIdx.x];
    ((u32x4*)(out + (size_t)slot * D))[threadIdx.x] = v;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// COLLECT = true is the collect tier's form of the same kernel: instead of keeping the best 8 rows of every half-list it
// keeps EVERY row whose score reaches the frame's fixed threshold thr_in[slot] (= the frame's exact k-th cosine so far minus
// the slack of its failed certificate, knn_rescore_kernel), up to 8 per half-list; a half-list that would need more marks
// the frame as overflowed (index -2 in its first entry).  Same tiles, same MFMA loop, same output layout.
// Seeded admission (both scoring kernels; the rationale and the protocol are written out above knn_score8_body): the blocks of
// library split s start their lists at the seeds the blocks of split s - 1 left for the same frames.
struct SeedArgs {
    float* tau;        // [frame slots] seeds, in the stage's score units; nullptr: no seeding in this launch
    int* flag;         // [splits][frame blocks] set once a block's seeds are visible
    int k;
    float margin;      // a seed = k-th best stage score seen so far - margin
    int* cnt;          // counter of the blocks that started from seeds
};

// one LOOK at the predecessor's flag; true: its seeds are visible to the plain loads that follow.  The result is BLOCK-uniform:
// thread 0 alone loads the flag (and runs the acquire), the verdict goes through LDS behind an unconditional barrier -- four
// waves looking for themselves could see the flag flip between their loads and fall one s_barrier out of step.
__device__ __forceinline__ bool seeds_look(const SeedArgs& sa, int split) {
    __shared__ int verdict[1];
    if (sa.tau == nullptr || split == 0) return false;                     // kernel arguments: uniform over the grid
    if (threadIdx.x == 0) {
        const int f = __hip_atomic_load(sa.flag + (size_t)(split - 1) * gridDim.x + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f != 0) {
            atomicAdd(sa.cnt, 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        verdict[0] = f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(verdict[0]) != 0;
}

// end of a block (after the barrier behind its last tile): thread = frame; the k-th largest score among the REAL entries of the
// frame's two half-lists (KH_ entries each, entry-major [KH_][512 lane-columns]) -> tau, then release + flag
// (WF_ frames per wave, FRAMES_ = 4 WF_ per block: 64 / 256 in the bf16 and the shipped fp8 kernel)
template <int KH_, int WF_ = 64, int FRAMES_ = 256>
__device__ __forceinline__ void seeds_publish(const SeedArgs& sa, const float* Lv, const int* Li, int64_t frame0, int split, bool seeded) {
    constexpr int LW_ = 4 * 2 * WF_;
    for (int col = threadIdx.x; col < FRAMES_; col += 256) {
        const int lcb = (col / WF_) * (2 * WF_) + ((col % WF_) >> 5) * 64 + (col & 31);
        float prev = INFINITY;
        for (int j = 0; j < sa.k; ++j) {
            float m = -INFINITY;
            for (int e = 0; e < 2 * KH_; ++e) {
                const int o = (e % KH_) * LW_ + lcb + (e / KH_) * 32;
                const float v = Lv[o];
                if (Li[o] >= 0 && v < prev) m = fmaxf(m, v);
            }
            prev = m;
        }
        float t = prev - sa.margin;                                    // -inf with fewer than k real entries
        if (seeded) {
            const float tin = sa.tau[frame0 + col];
            t = fmaxf(t, tin == tin ? tin : -INFINITY);
        }
        sa.tau[frame0 + col] = t;
    }
    // release: every storing wave's stores have left, then ONE agent-scope release and the flag (MI355X_MICROARCH.md)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(sa.flag + (size_t)split * gridDim.x + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool COLLECT>
__global__ __launch_bounds__(256, 1) void knn_score_kernel(const unsigned short* __restrict__ s_bf16,
                                                           const unsigned short* __restrict__ lib, int64_t M, int tiles_total,
                                                           int tiles_per_split, int P, float* __restrict__ cand_val,
                                                           int* __restrict__ cand_idx, const int* __restrict__ gate_cnt,
                                                           int gate_lo, int gate_hi, int by_count, const float* __restrict__ thr_in,
                                                           SeedArgs sa, int caps) {
    int n_slots = 0x7fffffff;
    {
        int c;
        if (!gate_open(gate_cnt, gate_lo, gate_hi, c)) return;                       // block-uniform
        if (by_count && (int64_t)blockIdx.x * 256 >= c) return;                      // compacted frames: only c of them
        if (by_count) n_slots = c;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Lv = (float*)(smem + 2 * ABUF);
    int* Li = (int*)(Lv + FT * KP);            // both entry-major: [KH][512 lane-columns] -> wave-wide conflict-free access

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int64_t frame0 = (int64_t)blockIdx.x * FT;
    const int split = blockIdx.y;
    const int tile_begin = split * tiles_per_split;
    int tile_end = tile_begin + tiles_per_split;
    if (tile_end > tiles_total) tile_end = tiles_total;

    const bool seeded = !COLLECT && seeds_look(sa, split);
    float seed[2] = {-INFINITY, -INFINITY};
    if (seeded) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const float t = sa.tau[frame0 + 64 * w + 32 * ni + lr];
            seed[ni] = t == t ? t : -INFINITY;
        }
    }
    // COLLECT (round 4): collected rows go straight to the block's own segment of the candidate arrays, cand[slot][split][caps],
    // through one LDS counter per frame (the two half-waves of a frame column share it): `caps` (64) rows per frame and library
    // split instead of 8 per half-list, so a cluster of a few dozen near-copies still fits and only more than that overflows
    int* cntF = (int*)Lv;                      // [FT] (the list area is not used by the collect form)
    if (COLLECT) {
        cntF[tid] = 0;
        for (int e = tid; e < FT * caps; e += 256) {
            const size_t o = ((size_t)(frame0 + e / caps) * P + split) * caps + e % caps;
            cand_val[o] = -INFINITY;
            cand_idx[o] = -1;
        }
    } else {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < KH; ++q) { Lv[q * 512 + w * 128 + ni * 64 + lane] = seed[ni]; Li[q * 512 + w * 128 + ni * 64 + lane] = -1; }
    }

    // DMA source of this lane inside a piece: row (lane>>3) of the wave's 8-row group, chunk (lane&7)^(lane>>3)
    const int dma_row = 8 * w + (lane >> 3);
    const int dma_chunk = (lane & 7) ^ (lane >> 3);
    auto issue_tile = [&](int tile, int buf) {
        const unsigned short* g = lib + ((size_t)tile * LT + dma_row) * D + dma_chunk * 8;
        unsigned char* l = smem + buf * ABUF + w * PIECE;
#pragma unroll
        for (int sg = 0; sg < D / 64; ++sg)
            __builtin_amdgcn_global_load_lds((gptr_t)(g + sg * 64), (lptr_t)(l + sg * 4 * PIECE), 16, 0, 0);
    };
    if (tile_begin < tile_end) issue_tile(tile_begin, 0);

    // stationary B fragments: frame = frame0 + 64 w + 32 ni + lr, k = 16 ks + 8 lh .. +7
    bf16x8 bq[2][NK16];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const unsigned short* fp = s_bf16 + (size_t)(frame0 + 64 * w + 32 * ni + lr) * D + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < NK16; ++ks) bq[ni][ks] = *(const bf16x8*)(fp + 16 * ks);
    }

    // A-fragment read addresses: row lr -> piece group q = lr>>3, line rr = lr&7; chunk ((ks&3)*2+lh) ^ rr
    const int rr = lr & 7;
    int a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a_off[j] = (lr >> 3) * PIECE + rr * 128 + ((((j * 2 + lh) ^ rr)) << 4);

    // lane-column id = w*128 + ni*64 + lane; entry e of its list lives at [e*512 + id]
    float thr[2] = {seed[0], seed[1]};
    if (COLLECT) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int64_t slot = frame0 + 64 * w + 32 * ni + lr;
            thr[ni] = slot < n_slots ? thr_in[slot] : INFINITY;        // padding slots collect nothing
        }
    }
    const int lc0 = w * 128 + lane;
    __syncthreads();                                   // lists initialised, tile_begin landed (vmcnt(0) + barrier)

    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int buf = (tile - tile_begin) & 1;
        // The 12 DMA pieces of the next tile are issued one per four k-steps INSIDE the MFMA stream (an LDS-DMA
        // costs the issuing wave ~60-180 cycles when issued back to back, almost nothing in an MFMA shadow).
        // On the last tile the same tile is fetched again into the idle buffer: no branch in the stream.
        const int next_tile = tile + 1 < tile_end ? tile + 1 : tile;
        const unsigned short* gnext = lib + ((size_t)next_tile * LT + dma_row) * D + dma_chunk * 8;
        unsigned char* lnext = smem + (buf ^ 1) * ABUF + w * PIECE;
        const unsigned char* Ab = smem + buf * ABUF;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        // A fragments are read PD k-steps ahead of their MFMAs (PD + 1 rotating registers).  A/B-tested on the MI355X:
        // PD 2 -> 4 is worth +3.8 % / +0.6 % (correlated / uncorrelated frames); deeper prefetch, exact group waits, issuing
        // the DMA pieces early and reading the AGPR-resident B fragments directly from an inline-asm MFMA all measure within
        // +-1 %, and the kernel without any candidate fold runs 1.40 PF: the loop sits at the MFMA rate the chip sustains
        // under this load.
        constexpr int PD = 4;
        bf16x8 a[PD + 1];
#pragma unroll
        for (int i = 0; i < PD; ++i) a[i] = *(const bf16x8*)(Ab + a_off[i & 3] + (i >> 2) * 4 * PIECE);
#pragma unroll
        for (int ks = 0; ks < NK16; ++ks) {
            if (ks + PD < NK16) a[(ks + PD) % (PD + 1)] = *(const bf16x8*)(Ab + a_off[(ks + PD) & 3] + ((ks + PD) >> 2) * 4 * PIECE);
            if ((ks & 3) == 1)
                __builtin_amdgcn_global_load_lds((gptr_t)(gnext + (ks >> 2) * 64), (lptr_t)(lnext + (ks >> 2) * 4 * PIECE), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);          // keep the read ahead of this step's MFMAs (hipcc otherwise
                                                        // sinks it next to its use and waits lgkmcnt(0) every step)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % (PD + 1)], bq[0][ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % (PD + 1)], bq[1][ks], acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- fold the 32 x 64 wave tile into the per-frame candidate lists ----
        // acc{ni}[r]: library row = row0 + (r&3) + 8*(r>>2) + 4*lh, frame column = 64 w + 32 ni + lr
        const int64_t row0 = (int64_t)tile * LT;
        const bool ragged = row0 + LT > M;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            f32x16& acc = ni == 0 ? acc0 : acc1;
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (ragged && row0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= M) acc[r] = -INFINITY;
                mx = fmaxf(mx, acc[r]);
            }
            if (COLLECT) {
                // every row at or above the fixed threshold is kept (rows, not a ranking: no minimum to maintain)
                // (mx > -inf: a threshold of -inf -- no finite k-th value -- must not keep the loop alive on retired entries)
                while (__builtin_amdgcn_ballot_w64(mx >= thr[ni] && mx > -INFINITY) != 0) {
                    const bool has = mx >= thr[ni] && mx > -INFINITY;
                    int rsel = 0;
#pragma unroll
                    for (int r = 1; r < 16; ++r) rsel = (acc[r] == mx) ? r : rsel;
                    if (has) {
                        const int fcol = 64 * w + 32 * ni + lr;
                        const int pos = atomicAdd(&cntF[fcol], 1);
                        if (pos < caps) {
                            const size_t o = ((size_t)(frame0 + fcol) * P + split) * caps + pos;
                            cand_val[o] = mx;
                            cand_idx[o] = (int)(row0 + (rsel & 3) + 8 * (rsel >> 2) + 4 * lh);
                        }
                    }
                    float nmx = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (has && r == rsel) acc[r] = -INFINITY;
                        nmx = fmaxf(nmx, acc[r]);
                    }
                    mx = nmx;
                }
                continue;
            }
            // Rare path (wave-uniform branch): some lane's best new score beats its list minimum.  The whole wave
            // runs it converged: every lane scans its own 8-entry list (entry-major LDS: conflict-free), lanes with a
            // candidate replace their minimum; repeated while any lane still holds a second candidate in this tile.
            while (__builtin_amdgcn_ballot_w64(mx > thr[ni]) != 0) {
                const bool has = mx > thr[ni];
                int rsel = 0;
#pragma unroll
                for (int r = 1; r < 16; ++r) rsel = (acc[r] == mx) ? r : rsel;
                float* lv = Lv + lc0 + ni * 64;
                int* li = Li + lc0 + ni * 64;
                float m1 = lv[0], m2 = INFINITY;          // smallest and second smallest entry
                int pos = 0;
#pragma unroll
                for (int e = 1; e < KH; ++e) {
                    float x = lv[e * 512];
                    bool lt = x < m1;
                    m2 = lt ? m1 : fminf(m2, x);
                    pos = lt ? e : pos;
                    m1 = lt ? x : m1;
                }
                if (has) {
                    lv[pos * 512] = mx;
                    li[pos * 512] = (int)(row0 + (rsel & 3) + 8 * (rsel >> 2) + 4 * lh);
                    thr[ni] = fminf(m2, mx);
                }
                // retire the inserted value and look for the next candidate of this lane
                float nmx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (has && r == rsel) acc[r] = -INFINITY;
                    nmx = fmaxf(nmx, acc[r]);
                }
                mx = nmx;
            }
        }
        __syncthreads();                               // next tile landed (vmcnt(0)), this buffer free for tile+2
    }

    if (COLLECT) {                                     // more rows above a frame's threshold than its segment holds: overflow mark
        if (cntF[tid] > caps) cand_idx[((size_t)(frame0 + tid) * P + split) * caps] = -2;
        return;
    }
    // ---- write this block's lists: cand[frame][P][KP]; entries 0..7 from the lower half-wave, 8..15 from the upper ----
    for (int e = tid; e < FT * KP; e += 256) {
        const int k = e % KP, col = e / KP;
        const int lc = (col >> 6) * 128 + ((col >> 5) & 1) * 64 + (k / KH) * 32 + (col & 31);
        const size_t o = (((size_t)(frame0 + col)) * P + split) * KP + k;
        cand_val[o] = Lv[(k % KH) * 512 + lc];
        cand_idx[o] = Li[(k % KH) * 512 + lc];
    }
    if (!COLLECT && sa.tau != nullptr) seeds_publish<KH>(sa, Lv, Li, frame0, split, seeded);
}

// ----------------------------------------------------------------------------------------------
// Round 4: the split-bf16 collect tier of the STRICT search (alive_knn_search_strict with a lo-plane library)
// ----------------------------------------------------------------------------------------------
// The strict search certifies a frame with a deterministic bound on |bf16 stage score - exact cosine| (~1.8e-3: Cauchy-Schwarz
// on two rounding-error vectors, 20 x the real error), and on a dense library that bound is wider than the neighbour gaps: 74 000
// of 172 800 frames failed it, 70 000 of them overflowed the collect tier (more rows inside the band than its lists hold) and
// went through the VALU exact scan -- 3.65 s per search.  This tier puts an MFMA pass with a TIGHT deterministic bound in front
// of that scan: both operands as two bf16 planes (hi = bf16(v), lo = bf16(v - hi)), score = q_hi.r_hi + q_lo.r_hi + q_hi.r_lo,
//   |score - q^.r^| <= ||q^ - q_hi - q_lo|| + ||r^ - r_hi - r_lo|| + ||q_lo|| ||r_lo|| + fp32 accumulation of 3 x 768 products
//                  <= 2^-18 + 2^-18 + 2^-18 (1 + 2^-8) + 2304 * u * 1.01,
// plus the rounding of the rescoring arithmetic itself (< 1e-5).  u is the unit roundoff of the additions INSIDE
// v_mfma_f32_32x32x16_bf16, which the ISA guide does not state: round-to-nearest (u = 2^-24) gives 1.5e-4 + 1e-5; the bound is
// sized for a TRUNCATING accumulate instead (u = 2^-23: 2.9e-4 + 1e-5), the worst an fp32 adder can do, so that the deterministic
// guarantee does not rest on an undocumented property of the matrix pipe: SPLIT_BOUND = 3.0e-4, a sixth of the single-plane bound
// (round 4 shipped 1.7e-4 on the round-to-nearest assumption; the wider band collects a few more rows per frame).  Every
// row whose split score reaches v_k - SPLIT_BOUND (v_k: the frame's k-th exact cosine so far, a lower bound of the true one) is
// collected and rescored exactly together with the frame's current top-k; only frames with more such rows than the lists hold
// (genuine near-ties: clusters of near-copies) go on to the exact scan.
// Structure of knn_score_kernel<true>, re-cut for two planes: a wave keeps 32 frames x 2 planes stationary (the same 384
// registers), a library tile is streamed as TWO LDS tiles -- its hi rows (two MFMAs per k-step: q_hi, q_lo) while the lo rows
// travel into the other buffer, then its lo rows (one MFMA per k-step: q_hi) while the next tile's hi rows travel.
constexpr int FT2 = 128;                                // frames per block
constexpr int SPLIT_LDS = 2 * ABUF + FT2 * 4;           // two tile buffers + one counter per frame
constexpr int SPLIT_ROWS = 256;                         // candidate entries per frame a bulk collect launch may fill (ws_layout: rows_b)
constexpr float SPLIT_BOUND = 3.0e-4f;

// lib_lo[m][d] = bf16(r^ - bf16(r^)), r^ = rows[m][d] / norms[m] exactly as lib_pack_kernel forms it; zero rows up to M_pad
__global__ __launch_bounds__(256) void lib_lo_kernel(const unsigned short* __restrict__ lib, const float* __restrict__ rows,
                                                     const float* __restrict__ norms, int64_t M, int64_t M_pad,
                                                     unsigned short* __restrict__ lo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M_pad * D) return;
    const int64_t m = i / D;
    unsigned short v = 0;
    if (m < M) {
        const float q = rows[i] / norms[m];
        v = f32_to_bf16_rn(q - __uint_as_float((unsigned)lib[i] << 16));
    }
    lo[i] = v;
}

// compacts both planes of the flagged frames: hi[slot] = s_bf16[list[slot]], lo[slot] = bf16(s_f32 - hi); zero rows up to the next 128
__global__ __launch_bounds__(128) void gather_frames_split_kernel(const unsigned short* __restrict__ s_bf16, const float* __restrict__ s_f32,
                                                                  const int* __restrict__ list, const int* __restrict__ cnt, int gate_lo,
                                                                  unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
    const int c = *cnt;
    const int slot = blockIdx.x;
    if (c <= gate_lo || slot >= (c + FT2 - 1) / FT2 * FT2) return;
    if (threadIdx.x >= 96) return;
    u32x4 h = {0u, 0u, 0u, 0u}, l = {0u, 0u, 0u, 0u};
    if (slot < c) {
        const size_t ft = (size_t)list[slot];
        h = ((const u32x4*)(s_bf16 + ft * D))[threadIdx.x];
        const f32x4* qp = (const f32x4*)(s_f32 + ft * D) + 2 * threadIdx.x;
        const f32x4 q0 = qp[0], q1 = qp[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = (e < 2 ? q0[2 * e] : q1[2 * e - 4]) - __uint_as_float(h[e] << 16);
            const float b = (e < 2 ? q0[2 * e + 1] : q1[2 * e - 3]) - __uint_as_float(h[e] & 0xffff0000u);
            l[e] = (unsigned)f32_to_bf16_rn(a) | ((unsigned)f32_to_bf16_rn(b) << 16);
        }
    }
    ((u32x4*)(hi + (size_t)slot * D))[threadIdx.x] = h;
    ((u32x4*)(lo + (size_t)slot * D))[threadIdx.x] = l;
}

__global__ __launch_bounds__(256, 1) void knn_collect_split_kernel(const unsigned short* __restrict__ s_hi, const unsigned short* __restrict__ s_lo,
                                                                   const unsigned short* __restrict__ lib_hi,
                                                                   const unsigned short* __restrict__ lib_lo, int64_t M, int tiles_total,
                                                                   int tiles_per_split, int P, int caps, float* __restrict__ cand_val,
                                                                   int* __restrict__ cand_idx, const int* __restrict__ cnt_ptr, int gate_lo,
                                                                   const float* __restrict__ thr_in) {
    const int n_slots = *cnt_ptr;
    if (n_slots <= gate_lo || (int64_t)blockIdx.x * FT2 >= n_slots) return;           // block-uniform
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // collected rows go straight to the block's own segment of the candidate arrays, cand[slot][split][caps], through one LDS
    // counter per frame (the two half-waves of a frame column share it): a list of `caps` (64) rows per frame and library split
    // instead of 8 per half-list, so a cluster of a few dozen near-copies still fits and only more than that reaches the exact scan
    int* cntF = (int*)(smem + 2 * ABUF);       // [FT2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int64_t frame0 = (int64_t)blockIdx.x * FT2;
    const int split = blockIdx.y;
    const int tile_begin = split * tiles_per_split;
    int tile_end = tile_begin + tiles_per_split;
    if (tile_end > tiles_total) tile_end = tiles_total;
    if (tid < FT2) cntF[tid] = 0;
    for (int e = tid; e < FT2 * caps; e += 256) {      // empty entries of this block's segment
        const size_t o = ((size_t)(frame0 + e / caps) * P + split) * caps + e % caps;
        cand_val[o] = -INFINITY;
        cand_idx[o] = -1;
    }

    const int dma_row = 8 * w + (lane >> 3);
    const int dma_chunk = (lane & 7) ^ (lane >> 3);
    auto dma_src = [&](const unsigned short* lib, int tile) { return lib + ((size_t)tile * LT + dma_row) * D + dma_chunk * 8; };
    if (tile_begin < tile_end) {
        const unsigned short* g = dma_src(lib_hi, tile_begin);
        unsigned char* l = smem + w * PIECE;
#pragma unroll
        for (int sg = 0; sg < D / 64; ++sg)
            __builtin_amdgcn_global_load_lds((gptr_t)(g + sg * 64), (lptr_t)(l + sg * 4 * PIECE), 16, 0, 0);
    }
    // stationary B fragments, both planes: frame = frame0 + 32 w + lr, k = 16 ks + 8 lh .. +7
    bf16x8 bq[2][NK16];
    {
        const size_t fo = (size_t)(frame0 + 32 * w + lr) * D + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < NK16; ++ks) {
            bq[0][ks] = *(const bf16x8*)(s_hi + fo + 16 * ks);
            bq[1][ks] = *(const bf16x8*)(s_lo + fo + 16 * ks);
        }
    }
    const int rr = lr & 7;
    int a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a_off[j] = (lr >> 3) * PIECE + rr * 128 + ((((j * 2 + lh) ^ rr)) << 4);
    const int64_t slot = frame0 + 32 * w + lr;
    const float thr = slot < n_slots ? thr_in[slot] : INFINITY;        // padding slots collect nothing
    const size_t seg = ((size_t)slot * P + split) * caps;
    __syncthreads();                                   // segment and counters initialised, the first hi tile landed (vmcnt(0) + barrier)

    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int next_tile = tile + 1 < tile_end ? tile + 1 : tile;       // last tile: the same rows once more into the idle buffer
        constexpr int PD = 4;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // ---- phase 0: the tile's hi rows (buffer 0) against both frame planes; its lo rows travel into buffer 1 ----
        {
            const unsigned short* gnext = dma_src(lib_lo, tile);
            unsigned char* lnext = smem + ABUF + w * PIECE;
            const unsigned char* Ab = smem;
            bf16x8 a[PD + 1];
#pragma unroll
            for (int i = 0; i < PD; ++i) a[i] = *(const bf16x8*)(Ab + a_off[i & 3] + (i >> 2) * 4 * PIECE);
#pragma unroll
            for (int ks = 0; ks < NK16; ++ks) {
                if (ks + PD < NK16) a[(ks + PD) % (PD + 1)] = *(const bf16x8*)(Ab + a_off[(ks + PD) & 3] + ((ks + PD) >> 2) * 4 * PIECE);
                if ((ks & 3) == 1)
                    __builtin_amdgcn_global_load_lds((gptr_t)(gnext + (ks >> 2) * 64), (lptr_t)(lnext + (ks >> 2) * 4 * PIECE), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % (PD + 1)], bq[1][ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % (PD + 1)], bq[0][ks], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                               // lo rows landed (vmcnt(0)), buffer 0 free
        // ---- phase 1: the tile's lo rows (buffer 1) against the frames' hi plane; the next tile's hi rows travel into buffer 0 ----
        {
            const unsigned short* gnext = dma_src(lib_hi, next_tile);
            unsigned char* lnext = smem + w * PIECE;
            const unsigned char* Ab = smem + ABUF;
            bf16x8 a[PD + 1];
#pragma unroll
            for (int i = 0; i < PD; ++i) a[i] = *(const bf16x8*)(Ab + a_off[i & 3] + (i >> 2) * 4 * PIECE);
#pragma unroll
            for (int ks = 0; ks < NK16; ++ks) {
                if (ks + PD < NK16) a[(ks + PD) % (PD + 1)] = *(const bf16x8*)(Ab + a_off[(ks + PD) & 3] + ((ks + PD) >> 2) * 4 * PIECE);
                if ((ks & 3) == 1)
                    __builtin_amdgcn_global_load_lds((gptr_t)(gnext + (ks >> 2) * 64), (lptr_t)(lnext + (ks >> 2) * 4 * PIECE), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % (PD + 1)], bq[0][ks], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- collect: acc[r] = library row row0 + (r & 3) + 8 (r >> 2) + 4 lh, frame column 32 w + lr ----
        const int64_t row0 = (int64_t)tile * LT;
        const bool ragged = row0 + LT > M;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (ragged && row0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= M) acc[r] = -INFINITY;
            mx = fmaxf(mx, acc[r]);
        }
        while (__builtin_amdgcn_ballot_w64(mx >= thr && mx > -INFINITY) != 0) {
            const bool has = mx >= thr && mx > -INFINITY;
            int rsel = 0;
#pragma unroll
            for (int r = 1; r < 16; ++r) rsel = (acc[r] == mx) ? r : rsel;
            if (has) {
                const int pos = atomicAdd(&cntF[32 * w + lr], 1);
                if (pos < caps) {
                    cand_val[seg + pos] = mx;
                    cand_idx[seg + pos] = (int)(row0 + (rsel & 3) + 8 * (rsel >> 2) + 4 * lh);
                }
            }
            float nmx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (has && r == rsel) acc[r] = -INFINITY;
                nmx = fmaxf(nmx, acc[r]);
            }
            mx = nmx;
        }
        __syncthreads();                               // next hi tile landed (vmcnt(0)), buffer 1 free
    }
    // a frame with more rows above its threshold than its segment holds: overflow mark in the first entry (-> exact scan)
    if (tid < FT2 && cntF[tid] > caps) cand_idx[((size_t)(frame0 + tid) * P + split) * caps] = -2;
}

// ----------------------------------------------------------------------------------------------
// scoring on the block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3 operands, scales 2^0)
// ----------------------------------------------------------------------------------------------
// The same kernel shape at twice the MFMA rate and half the operand bytes: a k-step is 64 features (one 32-byte fragment
// per lane = two ds_read_b128), a library tile is 32 rows x 768 B = 24 KB, a wave's stationary frame fragments take 192
// registers instead of 384.  fp8 scores carry ~20x the error of bf16 scores (sigma ~ 1.5e-3 .. 2.5e-3 in cosine against
// ~1e-4), so the candidate lists are twice as deep -- 16 per half-wave list, 32 per frame and library split -- and the exact
// fp32 rescoring behind it is unchanged: the fp8 pass only has to put the true neighbours among the candidates.
constexpr int KH8 = 16;                    // entries per half-list
constexpr int KP8 = 2 * KH8;               // candidates per frame and split
constexpr int NK64 = D / 64;               // 12 MFMA k-steps
constexpr int NPIECE8 = 4 * (D / 128);     // 4 row groups x 6 segments of 128 B
constexpr int ABUF8 = NPIECE8 * PIECE;     // 27648 B per tile buffer
// Column tiles (32 frames each) per wave: 2 in the fp8 kernel (64 stationary frames per wave, 192 registers); 3 in the fp6 kernel
// (round 5: 96 frames per wave in 216 registers, one A fragment feeding three MFMAs, 384 frames per block)
constexpr int FT6 = 4 * 32 * 3;            // frames per block of the fp6 kernel
constexpr int SCORE8_LDS = 2 * ABUF8 + FT * KP8 * 8;    // 120832 B
constexpr int SCORE6_LDS = 2 * ABUF8 + FT6 * KP8 * 8;   // 153600 B
typedef int v8i __attribute__((ext_vector_type(8)));

// Seeded admission (tau != nullptr; searches of >= SEED_MIN_FB frame blocks, where a library split fills at least two rounds of
// the chip, so that the blocks of split s start after those of split s - 1 have ended): a block leaves, per frame,
// tau = (k-th best fp8 score it has seen, over its own and the earlier splits) - SEED_MARGIN, and the block of the NEXT split
// of the same frames starts its lists AT tau instead of at -inf -- every entry holds tau with index -1, so rows below it
// are never admitted and the list floor the certificate reads is tau until real rows replace it.  Without a seed a split's
// lists warm up from nothing: 16 (1 + ln(n / 16)) admissions per half-list, and every admission in any wave stops the
// block's matrix pipes at the next barrier (ablation -DALIVE_KNN_ABL_RARE_W=1: the rare path in ONE wave of four costs
// what it costs in all four).  A row below tau has an fp8 score more than SEED_MARGIN under k rows already seen: it is
// "outside" in the certificate's sense, with tau as the bound on its score, exactly like a row under a full list's floor.
// The hand-off is one LOOK at the previous split's flag (no waiting: nothing can hang, an unfinished predecessor just means
// an unseeded block) behind the release / acquire pair of the guide; a stale or missing tau can only cost time, never a
// result: whatever seed a block used is IN its lists, and the certificate bounds the outside rows by it.
constexpr float SEED_MARGIN8 = 0.02f * F8_SCALE * F8_SCALE;      // cosine 0.02: certificate slack (>= 0.0105) + 4 sigma of the fp8 error
constexpr float SEED_MARGIN6 = 0.024f * F6_SCALE * F6_SCALE;     // fp6: slack >= 0.014 (7 x 2.0e-3) + 4 sigma of 1.8e-3, rounded up
constexpr float SEED_MARGIN16 = 2.5e-3f;                         // bf16 stage: its certificate's slack is ~7e-4 (7 sigma of ~1e-4)
constexpr float SEED_MARGIN16_STRICT = 4.5e-3f;                  // strict search: above the largest deterministic bound (2 x 2^-9 + 1e-4)
constexpr int SEED_MIN_FB = 512;
constexpr int SEED_MIN_FB6 = 300;                                // fp6 kernel, 384-frame blocks: a little over one round of the chip is enough --
                                                                 // the look at the predecessor's flag never waits, an unfinished one means an unseeded block
static float seed_margin8() {              // ALIVE_KNN_SEED_MARGIN (cosine units): experiments only
    static const float m = [] {
        const char* e = getenv("ALIVE_KNN_SEED_MARGIN");
        const float v = e ? (float)atof(e) : 0.0f;
        return v > 0.0f ? v * F8_SCALE * F8_SCALE : SEED_MARGIN8;
    }();
    return m;
}
static float seed_margin6() {
    static const float m = [] {
        const char* e = getenv("ALIVE_KNN_SEED_MARGIN");
        const float v = e ? (float)atof(e) : 0.0f;
        return v > 0.0f ? v * F6_SCALE * F6_SCALE : SEED_MARGIN6;
    }();
    return m;
}

template <int FMT, int NCT8>
__device__ __forceinline__ void knn_score8_body(const unsigned char* __restrict__ s_f8,
                                                const unsigned char* __restrict__ lib, int64_t M, int tiles_total,
                                                int tiles_per_split, int P, float* __restrict__ cand_val,
                                                int* __restrict__ cand_idx, const int* __restrict__ gate_cnt,
                                                int gate_lo, int gate_hi, SeedArgs sa) {
    {
        int c;
        if (!gate_open(gate_cnt, gate_lo, gate_hi, c)) return;                       // block-uniform
    }
    // FMT: operand format of both MFMA operands (0 = fp8 e4m3, 2 = fp6 e2m3); NCT8 column tiles of 32 frames per wave
    constexpr int WF8 = 32 * NCT8;             // frames per wave
    constexpr int FT8 = 4 * WF8;               // frames per block (256 / 384)
    constexpr int LW8 = 4 * 2 * WF8;           // lane-columns per list entry row (two half-waves per frame column)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Lv = (float*)(smem + 2 * ABUF8);
    int* Li = (int*)(Lv + FT8 * KP8);          // both entry-major: [KH8][LW8 lane-columns]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int64_t frame0 = (int64_t)blockIdx.x * FT8;
    const int split = blockIdx.y;
    const int tile_begin = split * tiles_per_split;
    int tile_end = tile_begin + tiles_per_split;
    if (tile_end > tiles_total) tile_end = tiles_total;

    const bool seeded = seeds_look(sa, split);               // block-uniform (see above)
    float seed[NCT8];
#pragma unroll
    for (int ni = 0; ni < NCT8; ++ni) seed[ni] = -INFINITY;
    if (seeded) {
#pragma unroll
        for (int ni = 0; ni < NCT8; ++ni) {
            const float t = sa.tau[frame0 + WF8 * w + 32 * ni + lr];
            seed[ni] = t == t ? t : -INFINITY;
        }
    }
#pragma unroll
    for (int ni = 0; ni < NCT8; ++ni)
#pragma unroll
        for (int q = 0; q < KH8; ++q) { Lv[q * LW8 + w * (2 * WF8) + ni * 64 + lane] = seed[ni]; Li[q * LW8 + w * (2 * WF8) + ni * 64 + lane] = -1; }

    // DMA: wave w copies rows 8 w .. 8 w + 7 of the tile, 6 pieces of 8 rows x 128 B; chunk order swizzled on the source
    const int dma_row = 8 * w + (lane >> 3);
    const int dma_chunk = (lane & 7) ^ (lane >> 3);
    auto issue_tile = [&](int tile, int buf) {
        const unsigned char* g = lib + ((size_t)tile * LT + dma_row) * D + dma_chunk * 16;
        unsigned char* l = smem + buf * ABUF8 + w * PIECE;
#pragma unroll
        for (int sg = 0; sg < D / 128; ++sg)
            __builtin_amdgcn_global_load_lds((gptr_t)(g + sg * 128), (lptr_t)(l + sg * 4 * PIECE), 16, 0, 0);
    };
    if (tile_begin < tile_end) issue_tile(tile_begin, 0);

    // stationary B fragments: frame = frame0 + 64 w + 32 ni + lr, features 64 ks + 32 lh .. + 31 (32 bytes)
    v8i bq[NCT8][NK64];
#pragma unroll
    for (int ni = 0; ni < NCT8; ++ni) {
        const unsigned char* fp = s_f8 + (size_t)(frame0 + WF8 * w + 32 * ni + lr) * D + 32 * lh;
#pragma unroll
        for (int ks = 0; ks < NK64; ++ks) {
            if constexpr (FMT == 2) {           // 24 of the 32 bytes carry the 32 six-bit codes: no 8-register temporaries
                const u32x4 lo = *(const u32x4*)(fp + 64 * ks);
                const uint2 hi = *(const uint2*)(fp + 64 * ks + 16);
                bq[ni][ks] = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi.x, (int)hi.y, 0, 0};
            } else {
                const u32x4 lo = *(const u32x4*)(fp + 64 * ks), hi = *(const u32x4*)(fp + 64 * ks + 16);
                bq[ni][ks] = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            }
        }
    }

    // A fragment of k-step ks: row lr, bytes 64 ks + 32 lh .. +31 = chunks c0 = 4 (ks & 1) + 2 lh and c0 + 1 of segment ks >> 1;
    // chunk c of line rr sits at position c ^ rr
    const int rr = lr & 7;
    int a_off[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) a_off[j][c] = (lr >> 3) * PIECE + rr * 128 + (((4 * j + 2 * lh + c) ^ rr) << 4);

    // Candidate lists: 16 entries per lane and column group in LDS, cached in registers as four quarters of four -- the
    // smallest value of each quarter (qv) and where it sits (qp).  The list minimum (the admission threshold) is min(qv); an
    // insertion overwrites that entry and rescans ONE quarter (4 LDS reads) instead of the list.
    float qv[NCT8][4];
    int qp[NCT8][4];
#pragma unroll
    for (int ni = 0; ni < NCT8; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) { qv[ni][q] = seed[ni]; qp[ni][q] = 4 * q; }
    float thr[NCT8];
#pragma unroll
    for (int ni = 0; ni < NCT8; ++ni) thr[ni] = seed[ni];
    const int lc0 = w * (2 * WF8) + lane;
    __syncthreads();

    auto load_a = [&](const unsigned char* Ab, int ks) {
        const unsigned char* q = Ab + (ks >> 1) * 4 * PIECE;
#ifndef ALIVE_KNN6_READ32
        if constexpr (FMT == 2) {
            // Only the 24 code bytes of the 32-byte slot leave the LDS (the kernel moves 120 KB through the LDS per 1 152 MFMA cycles:
            // 61.0 -> 57.7 ms per launch; -DALIVE_KNN6_READ32 restores the two b128 reads).  A compiler-visible ds_read_b64 of the tile
            // buffer makes hipcc wait vmcnt(0) for the LDS-DMA in flight, so the 8-byte half goes through asm -- issued
            // BEFORE the visible ds_read_b128 (whose address is tied to the asm): LDS reads of a wave return in order, so the wait
            // hipcc places in front of the fragment's first use (the MFMA takes both halves) covers the asm read too; an unknown
            // extra read in flight can only make hipcc's counted waits stricter.  tests/test_host_logic.py checks on the listing
            // that no instruction touches an asm read's registers before an lgkmcnt wait.
            unsigned off0 = (unsigned)(uintptr_t)(lptr_t)(q + a_off[ks & 1][0]);
            const unsigned off1 = (unsigned)(uintptr_t)(lptr_t)(q + a_off[ks & 1][1]);
            uint2 h8;
            asm volatile("ds_read_b64 %0, %2" : "=v"(h8), "+v"(off0) : "v"(off1));
            const u32x4 lo = *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)off0;
            return v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)h8.x, (int)h8.y, 0, 0};
        }
#endif
        const u32x4 lo = *(const u32x4*)(q + a_off[ks & 1][0]);
        const u32x4 hi = *(const u32x4*)(q + a_off[ks & 1][1]);
        return v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };

    // ---- folding a 32 x 32 accumulator tile (column group ni) into the per-frame candidate lists, in pieces ----
    // rows beyond M (only in the library's last tiles: a real, wave-uniform branch)
    auto mask_ragged = [&](f32x16& acc, int tile) {
        const int64_t row0 = (int64_t)tile * LT;
        if (row0 + LT > M) {
            asm volatile("" ::: "memory");
            const int left = (int)(M - row0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (4 * lh >= left - ((r & 3) + 8 * (r >> 2))) acc[r] = -INFINITY;
        }
    };
    // running maximum over 8 of the 16 registers: small enough to sit between two MFMAs of the next tile
    auto max8 = [&](const f32x16& acc, int lo, float m) {
#pragma unroll
        for (int r = 0; r < 8; ++r) m = fmaxf(m, acc[lo + r]);
        return m;
    };
    // the loop form on accumulators that already carry their register index: admits mx, then the largest score below it, ...
    // while any lane still holds an admitted score
    auto fold_loop = [&](f32x16& acc, int ni, int tile, float mx) {
        const int64_t row0 = (int64_t)tile * LT;
        float* lv = Lv + lc0 + ni * 64;
        int* li = Li + lc0 + ni * 64;
        while (__builtin_amdgcn_ballot_w64(mx > thr[ni]) != 0) {
            const bool has = mx > thr[ni];
            // quarter that holds the list minimum
            const bool b01 = qv[ni][1] < qv[ni][0], b23 = qv[ni][3] < qv[ni][2];
            const float m01 = b01 ? qv[ni][1] : qv[ni][0], m23 = b23 ? qv[ni][3] : qv[ni][2];
            const int p01 = b01 ? qp[ni][1] : qp[ni][0], p23 = b23 ? qp[ni][3] : qp[ni][2];
            const bool bq = m23 < m01;
            const int tpos = bq ? p23 : p01;                     // entry to overwrite
            const int tq = tpos >> 2;                            // its quarter
            if (has) {
                lv[tpos * LW8] = mx;
                li[tpos * LW8] = (int)(row0 + ((__float_as_uint(mx) & 3u) + 8u * ((__float_as_uint(mx) >> 2) & 3u)) + 4 * lh);
            }
            // rescan that quarter (after the write: LDS operations of a wave complete in order)
            const float* qb = lv + (tq * 4) * LW8;
            const float x0 = qb[0], x1 = qb[LW8], x2 = qb[2 * LW8], x3 = qb[3 * LW8];
            const bool c1 = x1 < x0, c3 = x3 < x2;
            const float n01 = c1 ? x1 : x0, n23 = c3 ? x3 : x2;
            const int e01 = c1 ? 1 : 0, e23 = c3 ? 3 : 2;
            const bool c = n23 < n01;
            const float nq = c ? n23 : n01;
            const int np = tq * 4 + (c ? e23 : e01);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool hit = has && tq == q;
                qv[ni][q] = hit ? nq : qv[ni][q];
                qp[ni][q] = hit ? np : qp[ni][q];
            }
            thr[ni] = fminf(fminf(qv[ni][0], qv[ni][1]), fminf(qv[ni][2], qv[ni][3]));
            // next candidate of this lane: the largest score below the one just taken (a lane that took none keeps its mx,
            // which may already be its 2nd or 3rd value -- it must not fall back to the top one)
            float nmx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) nmx = fmaxf(nmx, acc[r] < mx ? acc[r] : -INFINITY);
            mx = has ? nmx : mx;
        }
    };

    auto fold_rare = [&](f32x16& acc, int ni, int tile, float mx) {
        if (__builtin_amdgcn_ballot_w64(mx > thr[ni]) == 0) return;
#ifdef ALIVE_KNN_ABL_NORARE            // ablation build (tools/bench_knn.py): the fold's fast path only -- results are WRONG, timing only
        return;
#endif
#ifdef ALIVE_KNN_ABL_RARE_W            // ablation build: only waves below this number take the rare path (WRONG results): what a trip
        if (w >= ALIVE_KNN_ABL_RARE_W) return;      // costs the OTHER waves of the block through the per-tile barrier
#endif
        // Rare path (wave-uniform; by ablation ~1500 cycles with the matrix pipe idle, 15 % of the kernel on the bench batch and
        // 30 % on uncorrelated frames), cut for the common case of ONE admitted row per lane:
        //  * every lane's list minimum and its quarter are known from the register caches, so the quarter is read from LDS
        //    first and arrives under the packing / top-2 work below (the rescan used to wait behind the entry write);
        //  * the register index r goes into the 4 low mantissa bits of every score (2^-19 relative, far below the fp8 error):
        //    the 16 values of a lane are distinct and the row of a score is score & 15;
        //  * best and runner-up come out of one v_med3 / v_max chain; the runner-up (and whatever lies below it) is only
        //    looked at by the loop form when some lane's runner-up is admitted too.
        const int64_t row0 = (int64_t)tile * LT;
        float* lv = Lv + lc0 + ni * 64;
        int* li = Li + lc0 + ni * 64;
        const bool b01 = qv[ni][1] < qv[ni][0], b23 = qv[ni][3] < qv[ni][2];
        const float m01 = b01 ? qv[ni][1] : qv[ni][0], m23 = b23 ? qv[ni][3] : qv[ni][2];
        const int p01 = b01 ? qp[ni][1] : qp[ni][0], p23 = b23 ? qp[ni][3] : qp[ni][2];
        const int tpos = m23 < m01 ? p23 : p01;                 // entry to overwrite: the list minimum
        const int tq = tpos >> 2, te = tpos & 3;
        const float* qb = lv + (tq * 4) * LW8;
        float x0 = qb[0], x1 = qb[LW8], x2 = qb[2 * LW8], x3 = qb[3 * LW8];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = __uint_as_float((__float_as_uint(acc[r]) & ~15u) | (unsigned)r);
        if (row0 + LT > M) {                            // -inf with index bits is a NaN: put the rows beyond M back to -inf
            asm volatile("" ::: "memory");
            const int left = (int)(M - row0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (4 * lh >= left - ((r & 3) + 8 * (r >> 2))) acc[r] = -INFINITY;
        }
        float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            m2 = __builtin_amdgcn_fmed3f(m1, m2, acc[r]);       // m1 >= m2: the median is the new runner-up
            m1 = fmaxf(m1, acc[r]);
        }
        const bool has = m1 > thr[ni];
        if (has) {
            const unsigned u = __float_as_uint(m1);
            lv[tpos * LW8] = m1;
            li[tpos * LW8] = (int)(row0 + ((u & 3u) + 8u * ((u >> 2) & 3u)) + 4 * lh);
        }
        // the quarter as it stands after the write: the new entry in place of the old minimum
        x0 = (has && te == 0) ? m1 : x0;
        x1 = (has && te == 1) ? m1 : x1;
        x2 = (has && te == 2) ? m1 : x2;
        x3 = (has && te == 3) ? m1 : x3;
        const bool c1 = x1 < x0, c3 = x3 < x2;
        const float n01 = c1 ? x1 : x0, n23 = c3 ? x3 : x2;
        const int e01 = c1 ? 1 : 0, e23 = c3 ? 3 : 2;
        const bool c = n23 < n01;
        const float nq = c ? n23 : n01;
        const int np = tq * 4 + (c ? e23 : e01);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool hit = has && tq == q;
            qv[ni][q] = hit ? nq : qv[ni][q];
            qp[ni][q] = hit ? np : qp[ni][q];
        }
        thr[ni] = fminf(fminf(qv[ni][0], qv[ni][1]), fminf(qv[ni][2], qv[ni][3]));
        fold_loop(acc, ni, tile, m2);                           // returns at once unless some lane's runner-up is admitted too
    };
    // One tile: 24 MFMAs into (c0, c1).  The accumulators of the PREVIOUS tile (p0, p1) are folded in the shadow of this tile's
    // first MFMAs, a few VALU instructions after each (one wave per SIMD: nothing else would fill the matrix pipe between the
    // last MFMA of a tile and the end of its fold -- measured 23 ms of 107 with the fold behind the tile).  Before the first
    // tile p0 / p1 hold -inf: its fold finds nothing.
    if constexpr (NCT8 == 3) {
    // Three column tiles per wave (the fp6 kernel): 36 MFMAs per tile into (c0, c1, c2); the previous tile's accumulators
    // (p0, p1, p2) are copied out and folded in steps 6 .. 10.  Same pieces as the two-tile form below, one more of each.
    auto do_tile3 = [&](int tile, f32x16& c0, f32x16& c1, f32x16& c2, f32x16& p0, f32x16& p1, f32x16& p2) {
        const int buf = (tile - tile_begin) & 1;
        const int next_tile = tile + 1 < tile_end ? tile + 1 : tile;
        const unsigned char* gnext = lib + ((size_t)next_tile * LT + dma_row) * D + dma_chunk * 16;
        unsigned char* lnext = smem + (buf ^ 1) * ABUF8 + w * PIECE;
        const unsigned char* Ab = smem + buf * ABUF8;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[r] = 0.0f; c1[r] = 0.0f; c2[r] = 0.0f; }
#ifndef ALIVE_KNN6_PD
#define ALIVE_KNN6_PD 1                 // A fragments requested this many k-steps ahead of their MFMAs
#endif
        constexpr int PD6 = ALIVE_KNN6_PD, NA6 = PD6 + 1;
        // ablation switches (timing only, WRONG results: tools/ab_build.sh x.so knn.hip -DALIVE_KNN6_ABL=<bits>): 1 no LDS-DMA of the next
        // tile, 2 no fragment reads after the tile's first, 4 no barrier at the end of the tile
#ifndef ALIVE_KNN6_ABL
#define ALIVE_KNN6_ABL 0
#endif
        constexpr bool ABL_NODMA = (ALIVE_KNN6_ABL & 1) != 0, ABL_NOLDS = (ALIVE_KNN6_ABL & 2) != 0, ABL_NOBAR = (ALIVE_KNN6_ABL & 4) != 0;
        v8i a[NA6];
#pragma unroll
        for (int i = 0; i < PD6; ++i) a[i] = load_a(Ab, i);
        float pm0 = -INFINITY, pm1 = -INFINITY, pm2 = -INFINITY;
        ALIVE_CHAIN_GAP(7);
#define K8_STEP3(ks, AFTER0, AFTER1, AFTER2)                                                                                     \
        {                                                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(ks) % NA6], bq[0][ks], c0, FMT, FMT, 0, 127, 0, 127); \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            if ((ks) + PD6 < NK64 && !ABL_NOLDS) a[((ks) + PD6) % NA6] = load_a(Ab, (ks) + PD6);                                 \
            if ((ks) < D / 128 && !ABL_NODMA)                                                                                    \
                __builtin_amdgcn_global_load_lds((gptr_t)(gnext + (ks) * 128), (lptr_t)(lnext + (ks) * 4 * PIECE), 16, 0, 0);    \
            AFTER0;                                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(ks) % NA6], bq[1][ks], c1, FMT, FMT, 0, 127, 0, 127); \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            AFTER1;                                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(ks) % NA6], bq[2][ks], c2, FMT, FMT, 0, 127, 0, 127); \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            AFTER2;                                                                                                              \
        }
        f32x16 v0, v1, v2;
        auto acc_read = [&](const f32x16& a_, int lo, int hi, f32x16& v) {
#ifdef ALIVE_KNN_ABL_NOREAD            // ablation (timing only, WRONG results): what the 48 accumulator copies of a tile cost
#pragma unroll
            for (int r = lo; r < hi; ++r) v[r] = -INFINITY;
#else
#pragma unroll
            for (int r = lo; r < hi; ++r) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[r]) : "a"(a_[r]));
#endif
        };
        auto acc_pin = [&]() { asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2)); };
        K8_STEP3(0, (void)0, (void)0, (void)0) K8_STEP3(1, (void)0, (void)0, (void)0) K8_STEP3(2, (void)0, (void)0, (void)0)
        K8_STEP3(3, (void)0, (void)0, (void)0) K8_STEP3(4, (void)0, (void)0, (void)0) K8_STEP3(5, (void)0, (void)0, acc_pin())
        // (each column tile's copy is folded before the next one is read: one 16-register copy live at a time)
        K8_STEP3(6, acc_read(p0, 0, 8, v0), (acc_pin(), acc_read(p0, 8, 16, v0), mask_ragged(v0, tile - 1)), (acc_pin(), pm0 = max8(v0, 0, pm0)))
        K8_STEP3(7, pm0 = max8(v0, 8, pm0), fold_rare(v0, 0, tile - 1, pm0), (acc_pin(), acc_read(p1, 0, 8, v1)))
        K8_STEP3(8, (acc_pin(), acc_read(p1, 8, 16, v1), mask_ragged(v1, tile - 1)), (acc_pin(), pm1 = max8(v1, 0, pm1)), pm1 = max8(v1, 8, pm1))
        K8_STEP3(9, fold_rare(v1, 1, tile - 1, pm1), (acc_pin(), acc_read(p2, 0, 8, v2)), (acc_pin(), acc_read(p2, 8, 16, v2), mask_ragged(v2, tile - 1)))
        K8_STEP3(10, (acc_pin(), pm2 = max8(v2, 0, pm2)), pm2 = max8(v2, 8, pm2), fold_rare(v2, 2, tile - 1, pm2))
        K8_STEP3(11, (void)0, (void)0, (void)0)
#undef K8_STEP3
        acc_pin();
        if (!ABL_NOBAR) __syncthreads();
    };
    auto fold_now3 = [&](f32x16& p0, f32x16& p1, f32x16& p2, int tile) {
        mask_ragged(p0, tile);
        mask_ragged(p1, tile);
        mask_ragged(p2, tile);
        fold_rare(p0, 0, tile, max8(p0, 8, max8(p0, 0, -INFINITY)));
        fold_rare(p1, 1, tile, max8(p1, 8, max8(p1, 0, -INFINITY)));
        fold_rare(p2, 2, tile, max8(p2, 8, max8(p2, 0, -INFINITY)));
    };
    {
        f32x16 A0, A1, A2, B0, B1, B2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { B0[r] = -INFINITY; B1[r] = -INFINITY; B2[r] = -INFINITY; }
        int tile = tile_begin;
        for (; tile + 1 < tile_end; tile += 2) {
            do_tile3(tile, A0, A1, A2, B0, B1, B2);
            do_tile3(tile + 1, B0, B1, B2, A0, A1, A2);
        }
        if (tile < tile_end) {
            do_tile3(tile, A0, A1, A2, B0, B1, B2);
            fold_now3(A0, A1, A2, tile);
        } else if (tile_begin < tile_end) {
            fold_now3(B0, B1, B2, tile_end - 1);
        }
    }
    } else {
    auto do_tile = [&](int tile, f32x16& c0, f32x16& c1, f32x16& p0, f32x16& p1) {
        const int buf = (tile - tile_begin) & 1;
        const int next_tile = tile + 1 < tile_end ? tile + 1 : tile;
        const unsigned char* gnext = lib + ((size_t)next_tile * LT + dma_row) * D + dma_chunk * 16;
        unsigned char* lnext = smem + (buf ^ 1) * ABUF8 + w * PIECE;
        const unsigned char* Ab = smem + buf * ABUF8;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[r] = 0.0f; c1[r] = 0.0f; }
        // hipcc waits lgkmcnt(0) in front of every MFMA that takes an LDS fragment, i.e. for EVERYTHING in flight -- a
        // prefetch issued before the MFMAs of a step is waited for at once (measured: 46 % of the loop idle).  So the
        // fragment of step ks + 1 is requested between the two MFMAs of step ks: the wait then falls in front of the next
        // step's first MFMA, ~120 cycles later, behind this step's second MFMA in the matrix pipe.
        v8i a[2];
        a[0] = load_a(Ab, 0);
        float pm0 = -INFINITY, pm1 = -INFINITY;
#ifndef ALIVE_KNN8_NO_CHAIN_GAP
        // (p0, p1) are read in steps 6 .. 9, beside this tile's chains: the shape of the accumulation-chain hazard of the fused
        // FilterBlocks (common.h ALIVE_CHAIN_GAP).  No wrong score has been seen here (brute-force audits of every bench frame),
        // the eight idle wait states per 24-MFMA tile are the precaution the listing scan asks of every such chain switch.
        ALIVE_CHAIN_GAP(7);
#endif
        // written out step by step: with the folds inside a `for` hipcc gives up unrolling it and indexes bq dynamically (scratch)
#define K8_STEP(ks, AFTER0, AFTER1)                                                                                              \
        {                                                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(ks) & 1], bq[0][ks], c0, FMT, FMT, 0, 127, 0, 127);              \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            if ((ks) + 1 < NK64) a[((ks) + 1) & 1] = load_a(Ab, (ks) + 1);                                                       \
            if ((ks) < D / 128) /* the 6 DMA pieces of the next tile go out in the FIRST half of this one: the barrier at its */  \
                                /* end waits vmcnt(0), and a piece issued in the last step exposes a whole L2 / HBM round trip */ \
                __builtin_amdgcn_global_load_lds((gptr_t)(gnext + (ks) * 128), (lptr_t)(lnext + (ks) * 4 * PIECE), 16, 0, 0);    \
            AFTER0;                                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(ks) & 1], bq[1][ks], c1, FMT, FMT, 0, 127, 0, 127);              \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            AFTER1;                                                                                                              \
        }
#ifndef ALIVE_KNN_FOLD_EARLY
        // The previous tile's accumulators (p0, p1) STAY in their AGPR set until the second half of this tile and are copied out
        // there by explicit v_accvgpr_read, eight at a time behind the MFMAs of steps 6 .. 9 (the gaps of steps 0 .. 5 carry the
        // LDS-DMA pieces), each group pinned behind its MFMA by an empty asm that names the running accumulators.  Left to itself
        // hipcc keeps ONE accumulator set and copies it to VGPRs at the END of the tile that produced it -- 64 v_accvgpr_read and
        // 64 v_accvgpr_write per tile behind a drained matrix pipe (found from the listing after ablation builds had shown that
        // the always-on part of the fold, not the admissions, was the larger half of the fold's cost: 78.8 ms with no fold at
        // all, 86.1 with the always-on part only, 91.8 with admissions).  With the reads late and spread there are two accumulator
        // sets in the listing (a[0:31] / a[32:63]), no copy at the tile boundary, and the kernel runs 77.8 ms instead of 90.5 on the
        // bench batch (98.2 instead of 107.2 on uncorrelated frames): faster than the build without any fold, because the pins
        // also stop hipcc from sinking MFMA chains.  Placement matters: all 32 reads behind ONE pin 84.0 ms, reads in the
        // DMA-carrying gaps of steps 0 .. 1 87.3 ms.
        f32x16 v0, v1;
        auto acc_read = [&](const f32x16& a, int lo, int hi, f32x16& v) {
#pragma unroll
            for (int r = lo; r < hi; ++r) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[r]) : "a"(a[r]));
        };
        auto acc_pin = [&]() { asm volatile("" : "+a"(c0), "+a"(c1)); };
        K8_STEP(0, (void)0, (void)0) K8_STEP(1, (void)0, (void)0) K8_STEP(2, (void)0, (void)0) K8_STEP(3, (void)0, (void)0)
        K8_STEP(4, (void)0, (void)0) K8_STEP(5, (void)0, acc_pin())
        K8_STEP(6, acc_read(p0, 0, 8, v0), (acc_pin(), acc_read(p0, 8, 16, v0), mask_ragged(v0, tile - 1)))
        K8_STEP(7, (acc_pin(), pm0 = max8(v0, 0, pm0)), pm0 = max8(v0, 8, pm0))
        K8_STEP(8, (acc_pin(), acc_read(p1, 0, 8, v1)), (acc_pin(), acc_read(p1, 8, 16, v1), mask_ragged(v1, tile - 1)))
        K8_STEP(9, (acc_pin(), pm1 = max8(v1, 0, pm1)), pm1 = max8(v1, 8, pm1))
        K8_STEP(10, (void)0, fold_rare(v0, 0, tile - 1, pm0))
        K8_STEP(11, (void)0, fold_rare(v1, 1, tile - 1, pm1))
#else   // -DALIVE_KNN_FOLD_EARLY: the form of rounds 2 .. 3 (A/B builds): the fold reads p0 / p1 as plain values in steps 0 .. 3
        K8_STEP(0, (mask_ragged(p0, tile - 1), pm0 = max8(p0, 0, pm0)), pm0 = max8(p0, 8, pm0))
        K8_STEP(1, (mask_ragged(p1, tile - 1), pm1 = max8(p1, 0, pm1)), pm1 = max8(p1, 8, pm1))
        K8_STEP(2, (void)0, fold_rare(p0, 0, tile - 1, pm0))
        K8_STEP(3, (void)0, fold_rare(p1, 1, tile - 1, pm1))
        K8_STEP(4, (void)0, (void)0) K8_STEP(5, (void)0, (void)0) K8_STEP(6, (void)0, (void)0) K8_STEP(7, (void)0, (void)0)
        K8_STEP(8, (void)0, (void)0) K8_STEP(9, (void)0, (void)0) K8_STEP(10, (void)0, (void)0) K8_STEP(11, (void)0, (void)0)
#endif
#undef K8_STEP
        // The accumulators are next read by the fold inside the NEXT tile, and hipcc sinks the whole c1 chain down to that use:
        // eight dependent MFMAs back to back behind the barrier, their A fragments parked in AGPRs.  An opaque use pins both
        // chains to this tile, interleaved as written.
        asm volatile("" : "+a"(c0), "+a"(c1));
        __syncthreads();                   // next tile landed (vmcnt(0)), this buffer free for tile + 2
    };
    auto fold_now = [&](f32x16& p0, f32x16& p1, int tile) {          // the last tile of the split has no successor to hide behind
        mask_ragged(p0, tile);
        mask_ragged(p1, tile);
        fold_rare(p0, 0, tile, max8(p0, 8, max8(p0, 0, -INFINITY)));
        fold_rare(p1, 1, tile, max8(p1, 8, max8(p1, 0, -INFINITY)));
    };

    f32x16 A0, A1, B0, B1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { B0[r] = -INFINITY; B1[r] = -INFINITY; }
    int tile = tile_begin;
    for (; tile + 1 < tile_end; tile += 2) {
        do_tile(tile, A0, A1, B0, B1);
        do_tile(tile + 1, B0, B1, A0, A1);
    }
    if (tile < tile_end) {                 // odd count: one more tile into A, then its own fold
        do_tile(tile, A0, A1, B0, B1);
        fold_now(A0, A1, tile);
    } else if (tile_begin < tile_end) {    // even count: the last tile's accumulators are in B
        fold_now(B0, B1, tile_end - 1);
    }
    }
    __syncthreads();

    // ---- cand[frame][P][KP8]: entries 0 .. 15 from the lower half-wave, 16 .. 31 from the upper ----
    for (int e = tid; e < FT8 * KP8; e += 256) {
        const int k = e % KP8, col = e / KP8;
        const int lc = (col / WF8) * (2 * WF8) + ((col % WF8) >> 5) * 64 + (k / KH8) * 32 + (col & 31);
        const size_t o = (((size_t)(frame0 + col)) * P + split) * KP8 + k;
        cand_val[o] = Lv[(k % KH8) * LW8 + lc];
        cand_idx[o] = Li[(k % KH8) * LW8 + lc];
    }
    if (sa.tau != nullptr) seeds_publish<KH8, WF8, FT8>(sa, Lv, Li, frame0, split, seeded);
}

// two entry points of the same body, so that a kernel trace tells the pass over the batch from the 1 024-frame probe
__global__ __launch_bounds__(256, 1) void knn_score8_kernel(const unsigned char* __restrict__ s_f8, const unsigned char* __restrict__ lib,
                                                            int64_t M, int tiles_total, int tiles_per_split, int P,
                                                            float* __restrict__ cand_val, int* __restrict__ cand_idx,
                                                            const int* __restrict__ gate_cnt, int gate_lo, int gate_hi, SeedArgs sa) {
    knn_score8_body<0, 2>(s_f8, lib, M, tiles_total, tiles_per_split, P, cand_val, cand_idx, gate_cnt, gate_lo, gate_hi, sa);
}
__global__ __launch_bounds__(256, 1) void knn_probe8_kernel(const unsigned char* __restrict__ s_f8, const unsigned char* __restrict__ lib,
                                                            int64_t M, int tiles_total, int tiles_per_split, int P,
                                                            float* __restrict__ cand_val, int* __restrict__ cand_idx, const int* __restrict__ gate) {
    knn_score8_body<0, 2>(s_f8, lib, M, tiles_total, tiles_per_split, P, cand_val, cand_idx, gate, 0, 1, SeedArgs{nullptr, nullptr, 0, 0.0f, nullptr});
}
// the fp6 stage (round 5): e2m3 operands on both sides, three column tiles per wave
__global__ __launch_bounds__(256, 1) void knn_score6_kernel(const unsigned char* __restrict__ s_f6, const unsigned char* __restrict__ lib,
                                                            int64_t M, int tiles_total, int tiles_per_split, int P,
                                                            float* __restrict__ cand_val, int* __restrict__ cand_idx,
                                                            const int* __restrict__ gate_cnt, int gate_lo, int gate_hi, SeedArgs sa) {
    knn_score8_body<2, 3>(s_f6, lib, M, tiles_total, tiles_per_split, P, cand_val, cand_idx, gate_cnt, gate_lo, gate_hi, sa);
}
__global__ __launch_bounds__(256, 1) void knn_probe6_kernel(const unsigned char* __restrict__ s_f6, const unsigned char* __restrict__ lib,
                                                            int64_t M, int tiles_total, int tiles_per_split, int P,
                                                            float* __restrict__ cand_val, int* __restrict__ cand_idx, const int* __restrict__ gate) {
    knn_score8_body<2, 3>(s_f6, lib, M, tiles_total, tiles_per_split, P, cand_val, cand_idx, gate, 0, 1, SeedArgs{nullptr, nullptr, 0, 0.0f, nullptr});
}

// ----------------------------------------------------------------------------------------------
// exact fp32 rescoring + top-k  (one wave per frame)
// ----------------------------------------------------------------------------------------------
// Wave sums of G per-lane partials at once, bitwise equal to G calls of wave_sum (the xor tree 32, 16, 8, 4, 2, 1):
// in the first log2(G) levels a lane keeps half of its values and hands the other half to its partner (own + partner's,
// the operands of wave_sum's add at that level), so G values cost G - 1 + (6 - log2 G) shuffles instead of 6 G.
// Returns the total of frame t = lane / (64 / G) (the 64 / G lanes of that group all hold it).
template <int G>
__device__ __forceinline__ float wave_sum_frames(float (&d)[G], int lane) {
    if constexpr (G >= 16) {
        const bool up = lane & 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = (up ? d[8 + j] : d[j]) + __shfl_xor(up ? d[j] : d[8 + j], 32);
    }
    if constexpr (G >= 8) {
        constexpr int o = G >= 16 ? 16 : 32;
        const bool up = lane & o;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = (up ? d[4 + j] : d[j]) + __shfl_xor(up ? d[j] : d[4 + j], o);
    }
    if constexpr (G >= 4) {
        constexpr int o = G >= 16 ? 8 : (G >= 8 ? 16 : 32);
        const bool up = lane & o;
#pragma unroll
        for (int j = 0; j < 2; ++j) d[j] = (up ? d[2 + j] : d[j]) + __shfl_xor(up ? d[j] : d[2 + j], o);
    }
    if constexpr (G >= 2) {
        constexpr int o = G >= 16 ? 4 : (G >= 8 ? 8 : (G >= 4 ? 16 : 32));
        const bool up = lane & o;
        d[0] = (up ? d[1] : d[0]) + __shfl_xor(up ? d[0] : d[1], o);
    }
    float v = d[0];
#pragma unroll
    for (int o = 32 / G; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// (value desc, index asc) ordering, wave-wide argmax; returns the winning lane
__device__ __forceinline__ int wave_argbest(float v, int idx) {
    float bv = v;
    int bi = idx;
    int bl = threadIdx.x & 63;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o);
        int oi = __shfl_xor(bi, o);
        int ol = __shfl_xor(bl, o);
        bool take = (ov > bv) || (ov == bv && (unsigned)oi < (unsigned)bi) || (ov == bv && oi == bi && ol < bl);
        if (take) { bv = ov; bi = oi; bl = ol; }
    }
    return bl;
}

// Round 6: the kernel was the sum of three costs of about equal size that one wave per frame could not overlap -- ~7 000 vector
// instructions per frame (4 000 of them in a 64-step argmax selection over sixteen-wide register arrays of which a 160-candidate frame
// fills three, 2 400 in IEEE divisions), ~650 LDS permutes (every __shfl is a ds_bpermute_b32) and the 3-KB row gathers: 6.2 ms per
// 172 800 frames where the gathers alone are worth ~2.  Now:
//   * PER (template): candidates per lane, the size of the register arrays (R = P * kp <= 64 PER lanes);
//   * the selection is a THRESHOLD, not a ranking: 16 removal steps over the lanes' running maxima (DPP reductions, no LDS) give the
//     k-th and the 16th best stage score, everything at or above min(k-th - prune, 16th) is kept (the same set the ranking kept, plus
//     ties), compacted to one candidate per lane through 512 B of LDS per wave; only a frame with more than 64 such rows takes the old
//     ranking loop;
//   * divisions by the norm through div_by (see there), four candidates' wave sums through wave_sum_frames (7 permutes instead of 24,
//     bitwise the same sums), candidate broadcasts through v_readlane, the final top-k through DPP max / min.
// Results are bit for bit what they were (same products, same summation trees, same tie-breaks).
template <int PER>
__global__ __launch_bounds__(256) void knn_rescore_kernel(const float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                          int P, int kp, const float* __restrict__ s_f32,
                                                          const float* __restrict__ rows, const float* __restrict__ norms,
                                                          int64_t Tt, int64_t idx_base, int k, float* __restrict__ out_val,
                                                          int* __restrict__ out_idx, const int* __restrict__ frame_list,
                                                          const int* __restrict__ gate_cnt, int gate_lo, int gate_hi,
                                                          int* __restrict__ flag_list, int* __restrict__ flag_cnt, float zsig,
                                                          int list_len, float pre_scale, float sd_prior,
                                                          const float* __restrict__ det_q, const float* __restrict__ det_lib,
                                                          float* __restrict__ thr_list, int collect, float overflow_slack,
                                                          const unsigned char* __restrict__ force_fail = nullptr) {
    // force_fail (fp6 stage): frames whose fp6 image clipped an element (|x| 2^5 > 7.5) -- their stage scores are off by more than
    // any error statistic of their candidates can show, so they go to the next tier whatever the certificate says
    __shared__ float cs_v[4][64];          // compaction of the selected candidates, one row per wave
    __shared__ int cs_i[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t slot = (int64_t)blockIdx.x * 4 + wv;     // candidate lists are indexed by slot
    int gc;
    if (gate_lo == GATE_BOOL) {            // an on / off word (the probe's history gate), not a count of listed frames
        if (*gate_cnt == 0) return;
        gc = 0x7fffffff;
    } else if (!gate_open(gate_cnt, gate_lo, gate_hi, gc)) {
        return;
    }
    if (slot >= Tt || (frame_list != nullptr && slot >= gc)) return;
    const int64_t ft = frame_list != nullptr ? frame_list[slot] : slot;     // frame: source row and output row
    const int R = P * kp;                  // kp candidates per frame and split: KP (bf16 scoring) or KP8 (fp8 scoring)
    const float* cv = cand_val + (size_t)slot * R;
    const int* ci = cand_idx + (size_t)slot * R;
    // collect != 0: the candidates are the rows the collect tier kept (knn_score_kernel<true>: EVERY row at or above the
    // frame's threshold, so nothing is pruned and there is no certificate to compute); a frame whose rows did not fit -- an
    // overflowed half-list, or more than 64 rows together with its current top-k -- is flagged for the exact scan instead.
    const bool certify = flag_list != nullptr && !collect;
    bool overflow = false;
    float c_cut = -INFINITY;               // certificate: the best prefilter score a row OUTSIDE the rescored set can have
    float my_pre = 0.0f;                   // prefilter score of this lane's candidate

    // ---- candidate selection: this lane ends up with (at most) one candidate ----
    int my_idx = -1;
    // smallest entry of each FULL partial list (list_len consecutive entries = an aligned group of lanes: 16 behind the fp8 / fp6 stage,
    // 8 behind the bf16 stage; an empty slot holds -inf, or the seed of a seeded list): rows that never entered
    auto list_floor = [&](float v) { return list_len > 8 ? group_min<true>(v) : group_min<false>(v); };
    // Pruning (certified searches only): a candidate whose prefilter score lies more than 2 z sigma below the k-th best
    // prefilter score cannot reach the exact top-k unless two score errors beyond z sigma coincide; it is not rescored (a
    // 3-KB row gather saved: the kernel is bound by them) and counts, like every row that was never a candidate, into c_cut --
    // so the certificate below still decides whether the frame's result stands.  At least MIN_RESCORE candidates are kept for
    // the error statistics.
    constexpr int MIN_RESCORE = 16;
    // Strict mode (det_q != nullptr; bf16 stage only): the bound on |stage score - exact cosine| is DETERMINISTIC --
    // Cauchy-Schwarz on the two rounding-error vectors, || q^ - bf16(q^) || (this frame, measured) + || bf16(q^) || *
    // max_R || r^ - bf16(r^) || (this library, measured) + the fp32 accumulation error of 768 exact products (<= 768 * 2^-23
    // * sum |q r| <= 9.3e-5) + the rounding of the rescoring arithmetic itself (< 3e-6): no statistics, no assumption.
    const float det_bound = det_q != nullptr ? det_q[ft] + 1.004f * det_lib[0] + 1.0e-4f : 0.0f;
    const float prune = !certify ? INFINITY
                                 : (det_q != nullptr ? 2.0f * det_bound : 2.0f * zsig * sd_prior) / pre_scale;     // in prefilter-score units
    if constexpr (PER == 0) {              // R <= 64: one candidate per lane as it stands
        if (lane < R) my_idx = ci[lane];
        if (collect) {
            overflow = __builtin_amdgcn_ballot_w64(my_idx == -2) != 0;
            my_idx = my_idx < 0 ? -1 : my_idx;
        }
        if (certify) {
            // (the floor of a list counts its EMPTY entries too: -inf in an unseeded list -- not full, no floor -- and the seed in a
            // seeded one, below which the scoring kernel admitted nothing)
            c_cut = list_floor(lane < R ? cv[lane] : -INFINITY);
            my_pre = (lane < R && my_idx >= 0) ? cv[lane] : -INFINITY;
            // k-th and MIN_RESCORE-th largest prefilter score of the frame (wave-wide, by removal)
            float rest = my_pre, sk = -INFINITY, s16 = -INFINITY;
            for (int j = 0; j < MIN_RESCORE; ++j) {
                const float m = wave_max_u(rest);
                if (j == k - 1) sk = m;
                s16 = m;
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(rest == m && m > -INFINITY);
                if (hit == 0) break;                                   // fewer candidates than that
                if (lane == (int)__builtin_ctzll(hit)) rest = -INFINITY;
            }
            const float cut = fminf(sk - prune, s16);                  // keep everything down to the cut, and at least 16
            if (my_pre < cut) {
                c_cut = fmaxf(c_cut, my_pre);
                my_idx = -1;
            }
        }
    } else {
        // R = P * kp <= 64 PER candidates, PER per lane in registers
        float v[PER];
        int id[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int e = j * 64 + lane;
            const bool in = e < R;
            v[j] = in ? cv[e] : -INFINITY;
            id[j] = in ? ci[e] : -1;
            if (collect && id[j] == -2) overflow = true;
            if (certify) c_cut = fmaxf(c_cut, list_floor(v[j]));       // empty entries included: -inf, or the seed of a seeded list
            if (id[j] < 0) v[j] = -INFINITY;
        }
        // k-th and MIN_RESCORE-th best stage score of the frame, by removal from a working copy (one instance per step)
        float t[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) t[j] = v[j];
        float sk = -INFINITY, s16 = -INFINITY;
        for (int it = 0; it < MIN_RESCORE; ++it) {
            float lm = t[0];
#pragma unroll
            for (int j = 1; j < PER; ++j) lm = fmaxf(lm, t[j]);
            const float m = wave_max_u(lm);
            if (it == k - 1) sk = m;
            s16 = m;
            if (!(m > -INFINITY)) break;                               // wave-uniform: fewer candidates than that
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(lm == m);
            if (lane == (int)__builtin_ctzll(hit)) {
                bool done = false;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const bool here = !done && t[j] == m;
                    t[j] = here ? -INFINITY : t[j];
                    done = done || here;
                }
            }
        }
        const float cut = fminf(sk - prune, s16);                      // (-inf when fewer than MIN_RESCORE candidates exist, or nothing is pruned)
        bool sel[PER];
        int total = 0;
        unsigned long long msk[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            sel[j] = id[j] >= 0 && v[j] >= cut;
            msk[j] = __builtin_amdgcn_ballot_w64(sel[j]);
            total += (int)__builtin_popcountll(msk[j]);
        }
        if (total <= 64) {                                             // wave-uniform
            int base = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(msk[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)msk[j], 0u));
                if (sel[j]) {
                    cs_v[wv][pos] = v[j];
                    cs_i[wv][pos] = id[j];
                } else if (certify) {
                    c_cut = fmaxf(c_cut, v[j]);                        // a candidate the cut left behind
                }
                base += (int)__builtin_popcountll(msk[j]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // (one wave writes and reads its own row: LDS operations of a wave complete in order)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lane < total) {
                my_idx = cs_i[wv][lane];
                my_pre = cs_v[wv][lane];
            }
        } else {
            // more than 64 rows at or above the cut (clusters of near-copies; every collect launch of a crowded frame): the 64 best by
            // stage score, one argmax per step
            float skr = -INFINITY;
            for (int selr = 0; selr < 64; ++selr) {
                float bv = v[0];
                int bj = 0;
#pragma unroll
                for (int j = 1; j < PER; ++j)
                    if (v[j] > bv) { bv = v[j]; bj = j; }
                int bid = -1;
#pragma unroll
                for (int j = 0; j < PER; ++j)
                    if (j == bj) bid = id[j];
                const int win = wave_argbest(bv, bid);
                const int widx = __shfl(bid, win);
                const float wval = __shfl(bv, win);
                if (selr == k - 1) skr = wval;
                if (!(wval > -INFINITY)) break;                        // wave-uniform: no candidate left
                if (selr >= MIN_RESCORE && wval < skr - prune) break;  // wave-uniform: the rest stays in v[] and counts into c_cut
                if (lane == selr) { my_idx = (wval > -INFINITY) ? widx : -1; my_pre = wval; }
                if (lane == win) {
#pragma unroll
                    for (int j = 0; j < PER; ++j)
                        if (j == bj) v[j] = -INFINITY;
                }
            }
            if (certify) {                 // candidates the selection of 64 left behind
#pragma unroll
                for (int j = 0; j < PER; ++j) c_cut = fmaxf(c_cut, v[j]);
            }
            if (collect) {                 // collected rows that did not fit the 64 lanes: the frame goes to the exact scan
                float left = -INFINITY;
#pragma unroll
                for (int j = 0; j < PER; ++j) left = fmaxf(left, v[j]);
                overflow = __builtin_amdgcn_ballot_w64(overflow || left > -INFINITY) != 0;
            }
        }
        if (collect) overflow = __builtin_amdgcn_ballot_w64(overflow) != 0;
    }
    if (collect) {
        // the frame's current exact top-k joins the candidates (its rows lie at or above the threshold and are collected again
        // in all but a z-sigma tail of the statistical mode; joining them makes that tail harmless), without duplicates
        unsigned long long free_mask = ~__builtin_amdgcn_ballot_w64(my_idx >= 0);
        for (int j = 0; j < k; ++j) {
            const int g = out_idx[(size_t)ft * k + j];
            if (g < 0) continue;                                       // wave-uniform
            const int el = (int)(g - idx_base);
            if (__builtin_amdgcn_ballot_w64(my_idx == el) != 0) continue;
            if (free_mask == 0) { overflow = true; break; }
            const int l = (int)__builtin_ctzll(free_mask);
            if (lane == l) { my_idx = el; my_pre = INFINITY; }
            free_mask &= free_mask - 1;
        }
    }

    // ---- exact cosine: sum_d s_hat[d] * (row[d] / norm), lanes split d (3 x float4 each) ----
    const f32x4* sp = (const f32x4*)(s_f32 + (size_t)ft * D);
    f32x4 s0 = sp[lane], s1 = sp[lane + 64], s2 = sp[lane + 128];
    float my_score = -INFINITY;
    // four candidates per trip: their 3-KB rows are requested together (one wave per frame: a serial walk pays one
    // gather latency per candidate); lanes past the last candidate are not visited
    const unsigned long long have = __builtin_amdgcn_ballot_w64(my_idx >= 0);
    const int hi = have == 0 ? 0 : 64 - (int)__builtin_clzll(have);
    for (int c0 = 0; c0 < hi; c0 += 4) {
        int idx[4];
        bool any = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            idx[u] = __builtin_amdgcn_readlane(my_idx, c0 + u);
            any |= idx[u] >= 0;
        }
        if (!any) continue;                              // wave-uniform
        f32x4 r0[4], r1[4], r2[4];
        float nn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ii = idx[u] >= 0 ? idx[u] : 0;
            const f32x4* rp = (const f32x4*)(rows + (size_t)ii * D);
            r0[u] = rp[lane];
            r1[u] = rp[lane + 64];
            r2[u] = rp[lane + 128];
            nn[u] = norms[ii];
        }
        float d4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const NormDiv nd = norm_div(nn[u]);
            float p = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) p = fmaf(s0[j], div_by(r0[u][j], nd), p);
#pragma unroll
            for (int j = 0; j < 4; ++j) p = fmaf(s1[j], div_by(r1[u][j], nd), p);
#pragma unroll
            for (int j = 0; j < 4; ++j) p = fmaf(s2[j], div_by(r2[u][j], nd), p);
            d4[u] = p;
        }
        const int tot = __builtin_bit_cast(int, wave_sum_frames<4>(d4, lane));     // candidate u's sum in lanes 16 u .. 16 u + 15
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float p = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tot, 16 * u));
            if (idx[u] >= 0 && lane == c0 + u) my_score = p;
        }
    }

    // ---- certificate statistics: error of the prefilter score on the rescored candidates of THIS frame ----
    // The frame's candidates are the rows most like its true neighbours, so their errors have the scale that matters (the
    // error of a stage score grows with the score, and rows with a few dominant coordinates have errors of their own kind:
    // profiles/r03_knn_zstats.json) -- but they are a SELECTED sample: picking the highest stage scores favours positive
    // errors, so their mean may only ever RAISE the bound (err_mu <= 0: the systematic shrink of rounded products, ~0.5 % of
    // the score on fp8), the spread is taken about zero (RMS >= standard deviation), and twice the largest error seen is a
    // second floor for the slack (heavy single-term errors are bounded, not Gaussian).
    float err_mu = 0.0f, err_sd = 0.0f, err_max = 0.0f;
    if (certify) {
        const bool okc = my_idx >= 0 && my_score > -INFINITY;
        const float e = okc ? my_pre * pre_scale - my_score : 0.0f;
        float st3[4] = {okc ? 1.0f : 0.0f, e, e * e, 0.0f};
        // (three wave sums at once; each group of 16 lanes ends up with one of them)
        const int tot = __builtin_bit_cast(int, wave_sum_frames<4>(st3, lane));
        const float n = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tot, 0));
        const float se = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tot, 16));
        const float see = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tot, 32));
        err_mu = fminf(se / fmaxf(n, 1.0f), 0.0f);
        err_sd = fmaxf(sqrtf(see / fmaxf(n, 1.0f)), sd_prior);      // never below the stage's typical error
        err_max = wave_max_u(fabsf(e));
        c_cut = wave_max_u(c_cut);
    }

    // ---- exact top-k, descending, ties to the lower library index (then to the lower lane) ----
    float vk = -INFINITY;
    for (int j = 0; j < k; ++j) {
        const float wv_ = wave_max_u(my_score);
        const unsigned key = (my_score == wv_) ? (unsigned)(my_idx < 0 ? 0x7fffffff : my_idx) : 0xffffffffu;
        const unsigned wi_u = wave_min_u(key);
        const unsigned long long winm = __builtin_amdgcn_ballot_w64(key == wi_u && my_score == wv_);
        const int win = winm != 0 ? (int)__builtin_ctzll(winm) : 0;
        const int wi = (wi_u == 0x7fffffffu || wi_u == 0xffffffffu) ? -1 : (int)wi_u;
        if (lane == 0) {
            out_val[(size_t)ft * k + j] = wv_;
            out_idx[(size_t)ft * k + j] = (wi < 0 || !(wv_ > -INFINITY)) ? -1 : (int)(idx_base + wi);
        }
        if (lane == win) my_score = -INFINITY;
        vk = wv_;
    }
    // A row outside the rescored set has prefilter score <= c_cut, hence (prefilter = exact + error, error ~ (mu, sd) as
    // measured on this frame's own candidates) an exact score below c_cut - mu + z sd except in the z-sigma tail.  If the
    // k-th exact score does not clear that, the frame goes to the next tier: the bf16 candidate stage behind the fp8 one,
    // the exact scan behind the bf16 one (alive_knn_search_fp8 / alive_knn_search).
    const bool forced = force_fail != nullptr && force_fail[ft] != 0;
    if (certify && lane == 0 && (c_cut > -INFINITY || forced)) {
        const float bound = det_q != nullptr ? c_cut * pre_scale + det_bound
                                             : c_cut * pre_scale - err_mu + fmaxf(zsig * err_sd, 2.0f * err_max);
        if (!(vk > bound) || forced) {
            // the frame goes to the next tier; for the collect tier it takes along the threshold below which no row can
            // belong to its top-k: its k-th exact cosine so far minus the slack it was just tested with (stage-score units)
            const int pos = atomicAdd(flag_cnt, 1);
            flag_list[pos] = (int)ft;
            // (overflow_slack > 0: the next tier is the split-bf16 collect pass of the strict search, which collects above v_k minus
            //  ITS bound, not above v_k minus the slack this certificate was tested with)
            if (thr_list != nullptr) thr_list[pos] = overflow_slack > 0.0f ? vk - overflow_slack : (vk - (bound - c_cut * pre_scale)) / pre_scale;
        }
    }
    if (collect && lane == 0 && overflow) {
        // the frame's lists did not hold every row above its threshold: next tier.  Its k-th exact cosine so far (exact scores of real
        // rows: a lower bound of the true k-th) minus the next tier's slack is the threshold that tier collects above.
        const int pos = atomicAdd(flag_cnt, 1);
        flag_list[pos] = (int)ft;
        if (thr_list != nullptr) thr_list[pos] = vk - overflow_slack;
    }
}

// PER from the candidate count of a launch: R = P * kp <= 64 PER
template <typename... A>
static void rescore_launch(unsigned grid, hipStream_t s, const float* cv, const int* ci, int P, int kp, A... rest) {
    const int R = P * kp;
    if (R <= 64) knn_rescore_kernel<0><<<grid, 256, 0, s>>>(cv, ci, P, kp, rest...);
    else if (R <= 192) knn_rescore_kernel<3><<<grid, 256, 0, s>>>(cv, ci, P, kp, rest...);
    else if (R <= 256) knn_rescore_kernel<4><<<grid, 256, 0, s>>>(cv, ci, P, kp, rest...);
    else if (R <= 512) knn_rescore_kernel<8><<<grid, 256, 0, s>>>(cv, ci, P, kp, rest...);
    else knn_rescore_kernel<16><<<grid, 256, 0, s>>>(cv, ci, P, kp, rest...);
}

// ----------------------------------------------------------------------------------------------
// a handful of frames (streaming): exact fp32 scan, no candidate stage
// ----------------------------------------------------------------------------------------------
// With Tt * k <= 64 (a ring of 8 .. 16 frames at k = 4) the bf16 scoring kernel would run one 256-frame block that is
// 97 % padding, followed by a rescoring pass that walks 64 candidates per frame one after the other.  Here every wave
// streams its share of the fp32 rows once, scores each row against ALL frames with the arithmetic of
// knn_rescore_kernel (so the values are bitwise those of the batch path: divide by the norm, fmaf over d in the same
// lane partition, xor-tree wave sum) and keeps the exact top-k of every frame in ONE register pair per lane:
// lane = frame * k + slot, sorted descending, ties to the lower row.  A second kernel merges the per-wave lists.
// HBM-bound: M * 3 KB once (154 MB at 50 k vectors).
constexpr int64_t SCAN_ROWS_MAX = 262144; // beyond this the 3-KB fp32 rows cost more than the bf16 pass (1.5 KB) saves
constexpr int SCAN_WAVES = 4;           // waves per block
constexpr int SCAN_MAX_LISTS = 4096;    // total waves

// PF: the next row of a wave is requested before the current one is scored (64.3 -> 59.9 us at 50 k rows x 8 frames).  Two other
// forms were slower: the frames in registers (73.5 us: 96 more registers, fewer waves to hide the row loads behind) and the frames in
// LDS with the prefetch (72 us).  The kernel is a mix of HBM latency, 12 IEEE divisions and 8 x 12 fmas + a wave sum per row.
template <bool PF>
__global__ __launch_bounds__(64 * SCAN_WAVES) void knn_scan_kernel(const float* __restrict__ s_f32, const float* __restrict__ rows,
                                                                   const float* __restrict__ norms, int64_t M, int Tt, int k,
                                                                   float* __restrict__ part_val, int* __restrict__ part_idx) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = blockIdx.x * SCAN_WAVES + wv, nw = gridDim.x * SCAN_WAVES;
    const int my_t = lane / k;                       // lane = frame * k + slot (lanes >= Tt * k idle)
    const bool live = lane < Tt * k;
    float lv = -INFINITY;
    int li = 0x7fffffff;
    f32x4 n0 = {}, n1 = {}, n2 = {};
    float nnx = 1.0f;
    if (PF && gw < M) {
        const f32x4* rp = (const f32x4*)(rows + (size_t)gw * D);
        n0 = rp[lane]; n1 = rp[lane + 64]; n2 = rp[lane + 128];
        nnx = norms[gw];
    }
    for (int64_t r = gw; r < M; r += nw) {
        f32x4 q0, q1, q2;
        float nn;
        if (PF) {
            q0 = n0; q1 = n1; q2 = n2; nn = nnx;
            if (r + nw < M) {
                const f32x4* rp = (const f32x4*)(rows + (size_t)(r + nw) * D);
                n0 = rp[lane]; n1 = rp[lane + 64]; n2 = rp[lane + 128];
                nnx = norms[r + nw];
            }
        } else {
            const f32x4* rp = (const f32x4*)(rows + (size_t)r * D);
            nn = norms[r];
            q0 = rp[lane]; q1 = rp[lane + 64]; q2 = rp[lane + 128];
        }
        const NormDiv nd = norm_div(nn);
#pragma unroll
        for (int j = 0; j < 4; ++j) { q0[j] = div_by(q0[j], nd); q1[j] = div_by(q1[j], nd); q2[j] = div_by(q2[j], nd); }
        float p = -INFINITY;                             // the score of this lane's frame (the wave sum is lane-uniform)
        for (int t = 0; t < Tt; ++t) {
            const f32x4* sp = (const f32x4*)(s_f32 + (size_t)t * D);
            const f32x4 s0 = sp[lane], s1 = sp[lane + 64], s2 = sp[lane + 128];
            float d = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) d = fmaf(s0[j], q0[j], d);
#pragma unroll
            for (int j = 0; j < 4; ++j) d = fmaf(s1[j], q1[j], d);
#pragma unroll
            for (int j = 0; j < 4; ++j) d = fmaf(s2[j], q2[j], d);
            d = wave_sum(d);
            if (live && t == my_t) p = d;
        }
        // sorted insert inside the k lanes of this frame: entries that rank before (p, r) stay, the others move down one
        const bool before = lv > p || (lv == p && li < (int)r);
        const float up_v = __shfl_up(lv, 1);
        const int up_i = __shfl_up(li, 1);
        const bool up_before = (lane % k == 0) ? true : (up_v > p || (up_v == p && up_i < (int)r));
        if (live && !before) {
            lv = up_before ? p : up_v;
            li = up_before ? (int)r : up_i;
        }
    }
    part_val[(size_t)gw * 64 + lane] = lv;
    part_idx[(size_t)gw * 64 + lane] = li;
}

// one block per frame: merge nw partial lists of k entries -> exact top-k (descending, ties to the lower row).
// Every thread owns the lists of waves tid, tid + 256, ... (<= 16).  K4 (k <= 4, the streaming default): the thread loads its lists
// whole -- one 16-byte value and one 16-byte index vector per list, all in flight together -- and a round is a wave-level arg-best, one
// LDS exchange between the four waves and one barrier: no global load after the first.  (Until round 4 every thread re-read the
// heads of all its lists in every round and the block reduced through an LDS tree: 26 us of dependent L2 round trips for k = 4.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <bool K4>
__global__ __launch_bounds__(256) void knn_scan_merge_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx,
                                                             int nw, int k, int64_t idx_base, float* __restrict__ out_val,
                                                             int* __restrict__ out_idx) {
    constexpr int NL = SCAN_MAX_LISTS / 256;
    __shared__ float sv[2][4];
    __shared__ int si[2][4], sw[2][4];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int head[NL];
    f32x4 pv4[K4 ? NL : 1];
    i32x4 pi4[K4 ? NL : 1];
#pragma unroll
    for (int i = 0; i < NL; ++i) head[i] = 0;
    if constexpr (K4) {
        // lane = frame * k + slot in a wave's 64-entry list: the k entries of frame t start at t * k; k <= 4 entries are read as
        // scalars when k < 4 (the vector would cross into the next frame's entries, which is harmless but may be unaligned)
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int w = tid + 256 * i;
            if (k == 4) {
                const bool in = w < nw;
                pv4[i] = *(const f32x4*)(part_val + (size_t)(in ? w : 0) * 64 + t * 4);
                pi4[i] = *(const i32x4*)(part_idx + (size_t)(in ? w : 0) * 64 + t * 4);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = w < nw && e < k;
                    pv4[i][e] = in ? part_val[(size_t)w * 64 + t * k + e] : -INFINITY;
                    pi4[i][e] = in ? part_idx[(size_t)w * 64 + t * k + e] : 0x7fffffff;
                }
            }
        }
    }
    float bv;
    int bi, bl;
    auto rescan = [&]() {
        bv = -INFINITY; bi = 0x7fffffff; bl = -1;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int w = tid + 256 * i;
            if (w < nw && head[i] < k) {
                float v;
                int id;
                if constexpr (K4) {
                    v = head[i] == 0 ? pv4[i][0] : (head[i] == 1 ? pv4[i][1] : (head[i] == 2 ? pv4[i][2] : pv4[i][3]));
                    id = head[i] == 0 ? pi4[i][0] : (head[i] == 1 ? pi4[i][1] : (head[i] == 2 ? pi4[i][2] : pi4[i][3]));
                } else {
                    v = part_val[(size_t)w * 64 + t * k + head[i]];
                    id = part_idx[(size_t)w * 64 + t * k + head[i]];
                }
                if (v > bv || (v == bv && id < bi)) { bv = v; bi = id; bl = i; }
            }
        }
    };
    rescan();
    for (int j = 0; j < k; ++j) {
        const int par = j & 1;
        const int win = wave_argbest(bv, bi);           // (equal (value, row) pairs cannot meet: a row sits in exactly one list)
        const float wvv = __shfl(bv, win);
        const int wii = __shfl(bi, win);
        if (lane == 0) { sv[par][wv] = wvv; si[par][wv] = wii; sw[par][wv] = wv * 64 + win; }
        __syncthreads();
        float gv = sv[par][0];
        int gi = si[par][0], gt = sw[par][0];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float ov = sv[par][q];
            const int oi = si[par][q];
            if (ov > gv || (ov == gv && (unsigned)oi < (unsigned)gi)) { gv = ov; gi = oi; gt = sw[par][q]; }
        }
        if (tid == 0) {
            out_val[(size_t)t * k + j] = gv;
            out_idx[(size_t)t * k + j] = (gi == 0x7fffffff || !(gv > -INFINITY)) ? -1 : (int)(idx_base + gi);
        }
        if (gt == tid && bl >= 0) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
                if (i == bl) head[i]++;
            rescan();
        }
    }
}

// ----------------------------------------------------------------------------------------------
// merge shards + gather + mean + blend.  Block = 32 consecutive frames of one window.
// ----------------------------------------------------------------------------------------------
// NPER = candidates per lane in the merge: S * k <= 64 * NPER (2: the usual k <= 8 over up to 16 shards; 8: k up to 64)
template <int NPER, int KMAX>
__global__ __launch_bounds__(256) void knn_merge_gather_kernel(const float* __restrict__ cand_val,
                                                               const int* __restrict__ cand_idx, int S, int k, float alpha,
                                                               float one_minus, const float* __restrict__ rows, const float* __restrict__ src,
                                                               int T, int64_t Tt, float* __restrict__ out,
                                                               int* __restrict__ final_idx) {
    __shared__ int sel[32][KMAX];
    __shared__ float tile[32][65];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * 32;
    const int nf = (T - t0) < 32 ? (T - t0) : 32;

    // phase 1: each wave merges 8 frames; the S*k candidates of a frame sit NPER per lane
    for (int f = wv * 8; f < wv * 8 + 8; ++f) {
        if (f >= nf) break;
        const int64_t ft = (int64_t)n * T + t0 + f;
        float v[NPER];
        int id[NPER];
#pragma unroll
        for (int j = 0; j < NPER; ++j) {
            int e = j * 64 + lane;
            bool in = e < S * k;
            int s = in ? e / k : 0, kk = in ? e % k : 0;
            size_t o = ((size_t)s * Tt + ft) * k + kk;
            id[j] = in ? cand_idx[o] : -1;
            v[j] = (in && id[j] >= 0) ? cand_val[o] : -INFINITY;
        }
        if (S == 1) {
            // a single shard: its list IS the merged list (sorted, -1 behind the last valid entry)
            if (lane < k) {
                sel[f][lane] = id[0];
                if (final_idx != nullptr && blockIdx.z == 0) final_idx[(size_t)ft * k + lane] = id[0];
            }
            continue;
        }
        for (int j = 0; j < k; ++j) {
            float bv = v[0];
            int bi = id[0], bj = 0;
#pragma unroll
            for (int q = 1; q < NPER; ++q) {
                const bool better = (v[q] > bv) || (v[q] == bv && (unsigned)id[q] < (unsigned)bi);
                bv = better ? v[q] : bv;
                bi = better ? id[q] : bi;
                bj = better ? q : bj;
            }
            int win = wave_argbest(bv, bi < 0 ? 0x7fffffff : bi);
            int wi = __shfl(bi, win);
            if (lane == 0) {
                sel[f][j] = wi;
                if (final_idx != nullptr && blockIdx.z == 0) final_idx[(size_t)ft * k + j] = wi;
            }
            if (lane == win) {
#pragma unroll
                for (int q = 0; q < NPER; ++q)
                    if (q == bj) v[q] = -INFINITY;
            }
        }
    }
    __syncthreads();

    // phase 2: 64-feature slabs; gather (lane = feature), transpose through LDS, store (lane = frame).
    // gridDim.z == 1: this block walks all 12 slabs; gridDim.z == 12 (few frames: streaming): one slab per block
    const int d_begin = gridDim.z == 1 ? 0 : blockIdx.z * 64, d_end = gridDim.z == 1 ? D : d_begin + 64;
    for (int d0 = d_begin; d0 < d_end; d0 += 64) {
        for (int f = wv; f < nf; f += 4) {
            float acc = 0.0f;
            for (int j = 0; j < k; ++j) {
                int idx = sel[f][j];
                float r = (idx >= 0) ? rows[(size_t)idx * D + d0 + lane] : __builtin_nanf("");
                acc = (j == 0) ? r : acc + r;
            }
            tile[f][lane] = acc / (float)k;
        }
        __syncthreads();
        for (int e = tid; e < 64 * 32; e += 256) {
            int f = e & 31, d = e >> 5;
            if (f < nf) {
                size_t o = ((size_t)n * D + d0 + d) * T + t0 + f;
                float m = tile[f][d];
                out[o] = m * one_minus + src[o] * alpha;
            }
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------------
// exact tier: brute-force fp32 scan with the rescoring arithmetic -- any number of frames, k <= 64
// ----------------------------------------------------------------------------------------------
// The last tier of the search (frames whose bf16 candidate set could not be certified) and the whole search for
// k > 8 (deeper than a half-list of the candidate stages).  The streaming scan above, cut for frame lists of any
// length: frames go in groups of G (G * k <= 64 lanes hold the sorted lists), a work item is (group, row slice w of
// nw); a wave keeps the G normalised frames of its group in registers (12 floats per lane and frame), streams its
// rows once and scores each against all G frames with the arithmetic of knn_rescore_kernel -- bitwise the values the
// other tiers return.  The number of frames comes from a device counter, so the launch is unconditional: waves without
// an item return at once.  HBM-bound for G = 16 (3 GB of rows per group at 1 M vectors, shared through L2 / MALL by the
// groups in flight); a last resort, not a fast path.
constexpr int EX_BLOCKS = 2048;            // launch size (x 4 waves); items are distributed by a wave-stride loop
constexpr int EX_WAVES_TARGET = 16384;     // row slices per group are chosen so that about this many items exist
constexpr int EX_NW_MAX = 1024, EX_NW_MIN = 4;

__host__ __device__ inline int exact_group(int k) { return k <= 4 ? 16 : (k <= 8 ? 8 : (k <= 16 ? 4 : (k <= 32 ? 2 : 1))); }
__host__ __device__ inline int exact_nw(int64_t groups) {
    int64_t nw = groups > 0 ? EX_WAVES_TARGET / groups : EX_NW_MAX;
    nw = nw / 4 * 4;
    return (int)(nw < EX_NW_MIN ? EX_NW_MIN : (nw > EX_NW_MAX ? EX_NW_MAX : nw));
}
// lists the exact tier may need for `frames` frames: groups * nw(groups) <= max(EX_WAVES_TARGET, EX_NW_MIN * groups)
static size_t exact_lists(int64_t frames, int k) {
    const int64_t groups = (frames + exact_group(k) - 1) / exact_group(k);
    const int64_t a = EX_WAVES_TARGET, b = EX_NW_MIN * groups;
    return (size_t)(a > b ? a : b) + 4;
}

template <int G>
__global__ __launch_bounds__(256, 2) void knn_exact_kernel(const float* __restrict__ s_f32, const float* __restrict__ rows,
                                                        const float* __restrict__ norms, int64_t M, int64_t Tt, int k,
                                                        const int* __restrict__ frame_list, const int* __restrict__ cnt_ptr,
                                                        float* __restrict__ part_val, int* __restrict__ part_idx, int max_count) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t count = Tt;
    if (cnt_ptr != nullptr) { const int c = *cnt_ptr; count = c < Tt ? c : Tt; }
    if (count <= 0 || count > max_count) return;           // more frames than this launch is meant for: another tier has them
    const int64_t groups = (count + G - 1) / G;
    const int nw = exact_nw(groups);
    const int64_t items = groups * nw;
    const int my_t = lane / k;                                    // lists: lane = frame * k + slot
    const int src_lane = (my_t < G ? my_t : G - 1) * (64 / G);    // where wave_sum_frames leaves that frame's total
    for (int64_t item = (int64_t)blockIdx.x * 4 + wv; item < items; item += (int64_t)gridDim.x * 4) {
        const int64_t g = item / nw;
        const int w = (int)(item - g * nw);
        f32x4 s0[G], s1[G], s2[G];
#pragma unroll
        for (int t = 0; t < G; ++t) {
            int64_t slot = g * G + t;
            if (slot >= count) slot = g * G;                       // padding of the last group: a copy of its first frame
            const int64_t ft = frame_list != nullptr ? frame_list[slot] : slot;
            const f32x4* sp = (const f32x4*)(s_f32 + (size_t)ft * D);
            s0[t] = sp[lane]; s1[t] = sp[lane + 64]; s2[t] = sp[lane + 128];
        }
        const bool live = lane < G * k && g * G + my_t < count;
        float lv = -INFINITY;
        int li = 0x7fffffff;
        f32x4 q0, q1, q2;
        float nn = 1.0f;
        if (w < M) {
            const f32x4* rp = (const f32x4*)(rows + (size_t)w * D);
            q0 = rp[lane]; q1 = rp[lane + 64]; q2 = rp[lane + 128];
            nn = norms[w];
        }
        for (int64_t r = w; r < M; r += nw) {
            f32x4 n0 = q0, n1 = q1, n2 = q2;                       // the next row is requested before this one is scored
            float nnn = nn;                                        // (G = 16: no registers left for it; the SIMD's other wave covers the latency)
            if (G < 16 && r + nw < M) {
                const f32x4* rp = (const f32x4*)(rows + (size_t)(r + nw) * D);
                n0 = rp[lane]; n1 = rp[lane + 64]; n2 = rp[lane + 128];
                nnn = norms[r + nw];
            }
            const NormDiv nd = norm_div(nn);
#pragma unroll
            for (int j = 0; j < 4; ++j) { q0[j] = div_by(q0[j], nd); q1[j] = div_by(q1[j], nd); q2[j] = div_by(q2[j], nd); }
            float d[G];
#pragma unroll
            for (int t = 0; t < G; ++t) {
                float a = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) a = fmaf(s0[t][j], q0[j], a);
#pragma unroll
                for (int j = 0; j < 4; ++j) a = fmaf(s1[t][j], q1[j], a);
#pragma unroll
                for (int j = 0; j < 4; ++j) a = fmaf(s2[t][j], q2[j], a);
                d[t] = a;
            }
            const float tot = wave_sum_frames<G>(d, lane);
            const float p = __shfl(tot, src_lane);
            // sorted insert inside the k lanes of this frame: entries that rank before (p, r) stay, the others move down one
            const bool before = lv > p || (lv == p && li < (int)r);
            const float up_v = __shfl_up(lv, 1);
            const int up_i = __shfl_up(li, 1);
            const bool up_before = (lane % k == 0) ? true : (up_v > p || (up_v == p && up_i < (int)r));
            if (live && !before) {
                lv = up_before ? p : up_v;
                li = up_before ? (int)r : up_i;
            }
            if (G < 16) {
                q0 = n0; q1 = n1; q2 = n2; nn = nnn;
            } else if (r + nw < M) {
                const f32x4* rp = (const f32x4*)(rows + (size_t)(r + nw) * D);
                q0 = rp[lane]; q1 = rp[lane + 64]; q2 = rp[lane + 128];
                nn = norms[r + nw];
            }
        }
        part_val[(size_t)item * 64 + lane] = lv;
        part_idx[(size_t)item * 64 + lane] = li;
    }
}

// one block per frame (block-stride over the frame count): merge the nw partial lists of its group -> exact top-k
template <int G>
__global__ __launch_bounds__(256) void knn_exact_merge_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx,
                                                              int64_t Tt, int k, int64_t idx_base, const int* __restrict__ frame_list,
                                                              const int* __restrict__ cnt_ptr, float* __restrict__ out_val,
                                                              int* __restrict__ out_idx, int max_count) {
    __shared__ float sv[256];
    __shared__ int si[256], sw[256];
    const int tid = threadIdx.x;
    int64_t count = Tt;
    if (cnt_ptr != nullptr) { const int c = *cnt_ptr; count = c < Tt ? c : Tt; }
    if (count > max_count) return;
    const int64_t groups = (count + G - 1) / G;
    const int nw = exact_nw(groups);                     // <= 1024: up to four sorted lists per thread
    constexpr int NL = EX_NW_MAX / 256;
    for (int64_t slot = blockIdx.x; slot < count; slot += gridDim.x) {
        const int64_t g = slot / G;
        const int t = (int)(slot - g * G);
        const int64_t ft = frame_list != nullptr ? frame_list[slot] : slot;
        int head[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) head[i] = 0;
        for (int j = 0; j < k; ++j) {
            float bv = -INFINITY;
            int bi = 0x7fffffff, bl = -1;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int w = tid + 256 * i;
                if (w < nw && head[i] < k) {
                    const size_t o = ((size_t)g * nw + w) * 64 + t * k + head[i];
                    const float v = part_val[o];
                    const int id = part_idx[o];
                    if (v > bv || (v == bv && id < bi)) { bv = v; bi = id; bl = i; }
                }
            }
            sv[tid] = bv; si[tid] = bi; sw[tid] = tid;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if (tid < o) {
                    const float ov = sv[tid + o];
                    const int oi = si[tid + o];
                    if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) { sv[tid] = ov; si[tid] = oi; sw[tid] = sw[tid + o]; }
                }
                __syncthreads();
            }
            if (tid == 0) {
                out_val[(size_t)ft * k + j] = sv[0];
                out_idx[(size_t)ft * k + j] = (si[0] == 0x7fffffff || !(sv[0] > -INFINITY)) ? -1 : (int)(idx_base + si[0]);
            }
            if (sw[0] == tid && bl >= 0) {
#pragma unroll
                for (int i = 0; i < NL; ++i)
                    if (i == bl) head[i]++;
            }
            __syncthreads();
        }
    }
}

// ---- small device-side control kernels of the tiered search (no host sync anywhere) ----
// stats[]: see alive_knn_search_stats
constexpr int COLLECT_MIN = 256;          // frames failing the bf16 certificate: up to this many go straight to the exact scan
constexpr int RESEARCH_MIN = 64;          // frames failing the fp8 certificate: up to this many go straight to the exact scan (a bf16
                                          // pass for a handful of frames still computes whole 256-frame tiles: ~9 ms at 1 M rows
                                          // against ~0.75 ms per group of 64 / k frames of exact scan.  256 was tried: at k = 8 the 178
                                          // frames of the bench batch then cost 111.5 instead of 104.9 ms per search)
enum { ST_FLAG8 = 0, ST_FLAG16 = 1, ST_PROBE_N = 2, ST_PROBE_FAIL = 3, ST_MODE = 4, ST_FIRST = 5, ST_PROBE_CNT = 6, ST_TIER = 7, ST_FLAGC = 8, ST_SEEDED = 9, ST_SEEDED16 = 10, ST_RESEARCH_MIN = 11, ST_COLLECT_MIN = 12,
       ST_PROBE_SKIPPED = 13,
       // Probe history (round 6), kept in the CALLER'S workspace across calls and never reset by a search: the probe costs 1.8 ms per
       // 172 800-frame search and says the same thing every time a library is searched with frames of one kind.  After two consecutive
       // searches on this workspace (same library size, same frame count: ST_HIST_TAG) in which the probe chose the low-precision stage
       // and fewer than 5 % of the batch then failed its certificate, the probe runs on every 16th search only; any search that breaks
       // the pattern re-arms it.  A fresh or foreign workspace (tag mismatch) always probes.  The history changes which tier a frame
       // is answered by, never the answer.
       ST_HIST_TAG = 16, ST_HIST_STREAK = 17, ST_HIST_CALLS = 18, ST_PROBE_GATE = 19,
       ST_WORDS = 32 };
constexpr int PROBE_EVERY = 16, PROBE_STREAK = 2;
// ST_TIER: which path the last search on this workspace took (written by every path, so that the host never has to
// re-derive the dispatch): 1 = streaming scan, 2 = exact scan of every frame (k > 8), 3 = bf16 first, 4 = fp8 first
enum { TIER_SCAN = 1, TIER_EXACT_ALL = 2, TIER_BF16 = 3, TIER_FP8 = 4, TIER_FP6 = 5 };
__global__ void stats_init_kernel(int* __restrict__ stats, int tier, int hist_tag = 0) {
    // (the two tier limits are written out with the counters, so that a reader never repeats the constants)
    if (threadIdx.x < ST_HIST_TAG)
        stats[threadIdx.x] = threadIdx.x == ST_TIER ? tier
                           : threadIdx.x == ST_RESEARCH_MIN ? RESEARCH_MIN : threadIdx.x == ST_COLLECT_MIN ? COLLECT_MIN : 0;
    if (threadIdx.x == ST_HIST_TAG) {                 // one thread owns the history words (see the enum)
        const bool valid = hist_tag != 0 && stats[ST_HIST_TAG] == hist_tag;
        const int streak = valid ? stats[ST_HIST_STREAK] : 0, calls = valid ? stats[ST_HIST_CALLS] : 0;
        const bool skip = streak >= PROBE_STREAK && (calls % PROBE_EVERY) != 0;
        stats[ST_HIST_TAG] = hist_tag;
        stats[ST_HIST_STREAK] = streak;
        stats[ST_HIST_CALLS] = calls + 1;
        stats[ST_PROBE_GATE] = skip ? 0 : 1;
        stats[ST_PROBE_SKIPPED] = skip ? 1 : 0;
    }
}
// end of a low-precision-first search: did it go the way the probe (or the history) said it would?
__global__ void probe_history_kernel(int* __restrict__ stats, int64_t Tt) {
    const bool clean = stats[ST_MODE] == 0 && (int64_t)stats[ST_FLAG8] * 20 <= Tt &&
                       (stats[ST_PROBE_GATE] == 0 || (int64_t)stats[ST_PROBE_FAIL] * 4 <= (int64_t)stats[ST_PROBE_N]);
    stats[ST_HIST_STREAK] = clean ? stats[ST_HIST_STREAK] + 1 : 0;
}
__global__ __launch_bounds__(64) void probe_gather_kernel(const unsigned char* __restrict__ s_f8, int64_t Tt, int n, int n_pad,
                                                          unsigned char* __restrict__ out, int* __restrict__ plist, const int* __restrict__ gate) {
    if (*gate == 0) return;                            // the history says this search needs no probe
    const int slot = blockIdx.x;
    if (slot >= n_pad) return;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (slot < n) {
        const int64_t ft = (int64_t)slot * Tt / n;
        if (threadIdx.x < D / 16) v = ((const u32x4*)(s_f8 + (size_t)ft * D))[threadIdx.x];
        if (threadIdx.x == 0) plist[slot] = (int)ft;
    }
    if (threadIdx.x < D / 16) ((u32x4*)(out + (size_t)slot * D))[threadIdx.x] = v;
}

// mode = 1 (bf16 first) iff the fp8 certificate failed on more than `num / den` of the probe sample
__global__ void probe_decide_kernel(int* __restrict__ stats, int n, int num, int den) {
    if (stats[ST_PROBE_GATE] == 0) return;             // no probe ran: ST_MODE stays 0 (the low-precision stage first), ST_PROBE_N 0
    const int fail = stats[ST_PROBE_CNT];
    stats[ST_PROBE_N] = n;
    stats[ST_PROBE_FAIL] = fail;
    stats[ST_MODE] = ((int64_t)fail * den > (int64_t)n * num) ? 1 : 0;
}

// bf16-first mode: every frame goes to the bf16 tier (list = identity)
__global__ __launch_bounds__(256) void flag_all_kernel(int* __restrict__ list, int* __restrict__ cnt, int64_t Tt,
                                                       const int* __restrict__ gate_cnt, int gate_lo, int gate_hi) {
    int c;
    if (!gate_open(gate_cnt, gate_lo, gate_hi, c)) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < Tt) list[i] = (int)i;
    if (i == 0) *cnt = (int)Tt;
}

// ---- greedy de-duplication of a library (generate_voice_library.py --dedup): keep[i] = no KEPT earlier frame among its
// k nearest has cosine > threshold.  The sequential definition is a recursion over a DAG (edges point to lower indices),
// evaluated by passes: a frame is decided once all its earlier near neighbours are.  state: 0 undecided, 1 kept, 2 dropped;
// transitions are monotone, so in-place reads of a neighbour decided in the same pass are harmless.
__global__ __launch_bounds__(256) void dedup_pass_kernel(const float* __restrict__ val, const int* __restrict__ idx, int64_t M, int k,
                                                         float threshold, int* __restrict__ state, int* __restrict__ undecided) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M || state[i] != 0) return;
    bool any_kept = false, all_dropped = true;
    for (int q = 0; q < k; ++q) {
        const int j = idx[i * k + q];
        if (j < 0 || j >= i || !(val[i * k + q] > threshold)) continue;
        const int sj = ((volatile int*)state)[j];
        any_kept |= sj == 1;
        all_dropped &= sj == 2;
    }
    if (any_kept) state[i] = 2;
    else if (all_dropped) state[i] = 1;
    else atomicAdd(undecided, 1);
}

// optional instrumentation (the *_timed entry points): events recorded on the search stream around the first-stage scoring
// kernel only; passed down explicitly, no state in the library

struct SearchPlan {
    int64_t Tt, Tt_pad;
    int tiles_total, split, tiles_per_split, P;
};

// Grid = (Tt_pad / 256 frame blocks) x (split library ranges), one block per CU at a time (1 wave / SIMD).
// The split is chosen so that the last round of blocks over the 256 CUs is as full as possible.
static SearchPlan make_plan(int64_t Tt, int64_t M, int split_cap = MAX_SPLIT, int ft = FT) {
    SearchPlan p;
    p.Tt = Tt;
    p.Tt_pad = (Tt + ft - 1) / ft * ft;
    p.tiles_total = (int)(((M + TILE - 1) / TILE * TILE) / LT);
    const int64_t fb = p.Tt_pad / ft;
    int max_split = p.tiles_total / 8;                 // at least 8 tiles (256 rows) per block
    if (max_split > split_cap) max_split = split_cap;
    if (max_split < 1) max_split = 1;
    int best = 1;
    double best_eff = -1.0;
    auto eff_of = [&](int sp) {
        const int64_t blocks = fb * sp, rounds = (blocks + 255) / 256;
        return (double)blocks / (double)(rounds * 256);
    };
    for (int sp = 1; sp <= max_split; ++sp) {
        const double eff = eff_of(sp);
        if (eff > best_eff + 0.02) { best_eff = eff; best = sp; }
    }
    if (ft != FT) {
        // 384-frame blocks (the fp6 stage): the greedy rule above climbs to 25 splits for 450 frame blocks (0.9988 against 0.9766 at
        // 5) -- 800 candidates per frame for the rescoring and 24 cold list starts per frame block.  Here: the FEWEST splits within
        // 3 % of the best fill.
        double top = 0.0;
        for (int sp = 1; sp <= max_split; ++sp) top = eff_of(sp) > top ? eff_of(sp) : top;
        for (int sp = 1; sp <= max_split; ++sp)
            if (eff_of(sp) >= top - 0.03) { best = sp; break; }
    }
    p.split = best;
    p.tiles_per_split = (p.tiles_total + best - 1) / best;
    p.P = best;
    return p;
}

constexpr int MAX_SPLIT8 = 1024 / KP8;    // the rescoring kernel takes up to 1024 candidates per frame
constexpr int FCAP = 16384;               // tier 1 of the bf16 re-search: up to this many flagged frames ...
constexpr int TIER1A = 256;               // tier 1a: up to one frame block of flagged frames, library cut MAX_SPLIT ways
constexpr int FPLAN = 4096;               // ... with the library split chosen for this many (blocks past the count exit at once)
constexpr float CERT_Z = 7.0f;            // sigmas of candidate-score error the certificates allow for
constexpr float SD_PRIOR6 = 2.0e-3f;      // fp6 stage: 1.8e-3 simulated / measured on unit vectors of Gaussian elements
constexpr float SD_PRIOR8R = 2.5e-4f;     // the fp8 stage on ROTATED operands (rot_codes_kernel): 2.2e-4 on a frame's true neighbours, 2.4e-4 over all pairs of the dense bank with the leading coordinate centred (profiles/r06_knn_pca_probe.json); a floor of the per-frame estimate, as for the other stages
constexpr float SD_PRIOR8 = 1.5e-3f;      // typical error (cosine units) of an fp8 / a bf16 candidate score: floor of the per-frame estimate
constexpr float SD_PRIOR16 = 8.0e-5f;
constexpr int PROBE_N = 1024;             // frames of the adaptive probe (fp8 searches of >= PROBE_MIN_T frames)
constexpr int64_t PROBE_MIN_T = 16384;
constexpr int PROBE_NUM = 11, PROBE_DEN = 20; // bf16 first when more than 55 % of the sample fail the fp8 certificate: fp8 + bf16 on a
                                              // fraction f costs t8 + f t16 / eff against t16 = 2.6 t8 (80 against 206 ms since the fp8
                                              // kernel's late accumulator copies; it was 40 % at t16 = 2.3 t8).  Measured on the dense
                                              // library, 49 % failing: fp8 first 216.9 ms per search, bf16 first 227.9 ms

// seeds for a launch of `fb` frame blocks x `split` library splits (nullptr members: no seeding); zeroes the flags on the stream
static SeedArgs seeds_for(const struct SearchWs& w, int64_t fb, int split, int k, float margin, int counter, hipStream_t s, int min_fb = SEED_MIN_FB);

// ---- workspace: ONE layout function for the size query, the searches and the stats pointer ----
struct SearchWs {
    SearchPlan p16, p8, pt, pp;            // bf16 main / fp8 main / tier-1 re-search / probe
    SearchPlan p6, pp6;                    // fp6 main / probe (384-frame blocks)
    SearchPlan pa;                         // tier 1a of the bf16 re-search: ONE frame block, as many library splits as the rescoring takes
    int fcap, probe_n, probe_pad, probe_pad6;
    float* s_f32; unsigned short* s_bf16; unsigned char* s_f8;
    float* cv; int* ci;                    // candidate lists of the main pass (and of tier 2 of the re-search)
    float* pv; int* pi;                    // partial lists: streaming scan / exact tier
    int* stats; int* list0; int* list1;    // counters; frames flagged by the fp8 / by the bf16 certificate
    unsigned short* s_c;                   // compacted bf16 rows of the frames being re-searched
    float* cv1; int* ci1;                  // tier-1 candidate lists
    unsigned char* s_p8; float* cvp; int* cip; int* p_list;   // probe: sample rows, lists, [frame of slot | flagged frames]
    float* thr1; int* list2;               // collect tier: thresholds of the frames in list1; frames that overflowed it
    float* dq;                             // || q^ - bf16(q^) || per frame (strict certificate)
    unsigned char* clip6;                  // fp6 stage: 1 = the frame's fp6 image clipped an element (-> it fails the certificate)
    unsigned short* s_c2h; unsigned short* s_c2l;   // split tier (strict search): both planes of the frames that failed the deterministic certificate
    float* c2v; int* c2i;                  // candidate lists of the collect tiers: [slot][split][caps]
    int rows_t, rows_b;                    // rows per frame they may fill: tier-1 launches (<= fcap frames) / bulk launches
    const unsigned short* lib_lo;          // set by the strict search when the library carries its lo plane
    float* tau; int* tau_flag;             // fp8 stage: per-frame admission seeds handed from split to split, and the hand-off flags
    const float* det_q; const float* det_lib;   // set by the strict search only: frame / library share of the deterministic bound
    size_t bytes;
};

static SearchWs ws_layout(void* base, int64_t Tt, int64_t M, int k, bool split = false) {
    SearchWs w;
    w.p16 = make_plan(Tt, M);
    w.p8 = make_plan(Tt, M, MAX_SPLIT8);
    w.p6 = make_plan(Tt, M, MAX_SPLIT8, FT6);
    const int64_t Tp = w.p16.Tt_pad;
    const int64_t Tp8 = w.p6.Tt_pad > Tp ? w.p6.Tt_pad : Tp;      // (the fp6 stage pads the frames to blocks of 384)
    w.fcap = Tp < FCAP ? (int)Tp : FCAP;
    w.pt = make_plan(w.fcap < FPLAN ? w.fcap : FPLAN, M);
    w.pa = make_plan(FT, M);
    w.probe_n = Tt >= PROBE_MIN_T ? PROBE_N : 0;
    w.probe_pad = (w.probe_n + FT - 1) / FT * FT;
    w.probe_pad6 = (w.probe_n + FT6 - 1) / FT6 * FT6;
    w.pp = make_plan(w.probe_n > 0 ? w.probe_n : 1, M, MAX_SPLIT8);
    w.pp6 = make_plan(w.probe_n > 0 ? w.probe_n : 1, M, MAX_SPLIT8, FT6);
    Arena a(base);
    w.stats = a.take<int>(ST_WORDS);       // first: its address must not depend on k (alive_knn_search_stats has no k)
    w.s_f32 = a.take<float>((size_t)Tt * D);
    w.s_bf16 = a.take<unsigned short>((size_t)Tp * D);
    w.s_f8 = a.take<unsigned char>((size_t)Tp8 * D);
    const size_t c8a = (size_t)Tp * w.p8.P * KP8, c8b = (size_t)w.p6.Tt_pad * w.p6.P * KP8;
    const size_t c16 = (size_t)Tp * w.p16.P * KP, c8 = c8a > c8b ? c8a : c8b;
    w.cv = a.take<float>(c16 > c8 ? c16 : c8);
    w.ci = a.take<int>(c16 > c8 ? c16 : c8);
    size_t lists = exact_lists(Tt, k <= ALIVE_MAX_K && k >= 1 ? k : 4);
    if (lists < (size_t)SCAN_MAX_LISTS) lists = SCAN_MAX_LISTS;
    w.pv = a.take<float>(lists * 64);
    w.pi = a.take<int>(lists * 64);
    w.list0 = a.take<int>((size_t)Tp);
    w.list1 = a.take<int>((size_t)Tp);
    w.s_c = a.take<unsigned short>((size_t)Tp * D);
    const size_t n1a = (size_t)(w.fcap < TIER1A ? w.fcap : TIER1A) * w.pa.P * KP, n1 = (size_t)w.fcap * w.pt.P * KP;
    w.cv1 = a.take<float>(n1 > n1a ? n1 : n1a);
    w.ci1 = a.take<int>(n1 > n1a ? n1 : n1a);
    const size_t ppad = w.probe_pad6 > w.probe_pad ? w.probe_pad6 : w.probe_pad;
    const size_t pcand = ppad * (size_t)(w.pp6.P > w.pp.P ? w.pp6.P : w.pp.P) * KP8;
    w.s_p8 = a.take<unsigned char>(ppad * D + 16);
    w.cvp = a.take<float>(pcand + 4);
    w.cip = a.take<int>(pcand + 4);
    w.p_list = a.take<int>(2 * ppad + 8);
    w.thr1 = a.take<float>((size_t)Tp);
    w.list2 = a.take<int>((size_t)Tp);
    w.dq = a.take<float>((size_t)Tp);
    w.clip6 = a.take<unsigned char>((size_t)Tp8);
    // rows per frame the collect tiers may fill (the rescoring kernel takes up to 1024): 1024 for up to FCAP frames (few frames: many
    // library splits, and a cluster must still fit one split's segment), SPLIT_ROWS for the bulk launches of a big batch
    const size_t c2n = (size_t)(Tp < FCAP ? Tp : FCAP) * 1024 > (size_t)Tp * SPLIT_ROWS ? (size_t)(Tp < FCAP ? Tp : FCAP) * 1024 : (size_t)Tp * SPLIT_ROWS;
    w.c2v = a.take<float>(c2n);
    w.c2i = a.take<int>(c2n);
    w.rows_t = (int)(c2n / (size_t)w.fcap < 1024 ? c2n / (size_t)w.fcap : 1024);
    w.rows_b = (int)(c2n / (size_t)Tp < 1024 ? c2n / (size_t)Tp : 1024);
    w.lib_lo = nullptr;
    w.tau = a.take<float>((size_t)Tp8);
    w.tau_flag = a.take<int>((size_t)(Tp / FT) * (MAX_SPLIT > MAX_SPLIT8 ? MAX_SPLIT : MAX_SPLIT8));
    w.det_q = nullptr;
    w.det_lib = nullptr;
    // the two frame planes of the split tier (1.5 KB per frame each) exist only in the strict search's workspace
    // (alive_knn_workspace_bytes_strict); they come last, so every other address is the same in both layouts
    w.s_c2h = split ? a.take<unsigned short>((size_t)Tp * D) : nullptr;
    w.s_c2l = split ? a.take<unsigned short>((size_t)Tp * D) : nullptr;
    w.bytes = a.used() + 1024;
    return w;
}

static SeedArgs seeds_for(const SearchWs& w, int64_t fb, int split, int k, float margin, int counter, hipStream_t s, int min_fb) {
    if (fb < min_fb || split < 2) return SeedArgs{nullptr, nullptr, 0, 0.0f, nullptr};
    (void)hipMemsetAsync(w.tau_flag, 0, (size_t)fb * split * sizeof(int), s);
    return SeedArgs{w.tau, w.tau_flag, k, margin, w.stats + counter};
}

}  // namespace

extern "C" int64_t alive_library_padded_rows(int64_t M) { return (M + TILE - 1) / TILE * TILE; }

extern "C" int alive_library_pack(const float* tokens, int64_t M, int Dd, void* lib_bf16, float* rows_f32, float* norms,
                                  void* stream) {
    ALIVE_CHECK_ARG(tokens && lib_bf16 && rows_f32 && norms, "alive_library_pack: null pointer");
    ALIVE_CHECK_ARG(Dd == D, "alive_library_pack: feature dim %d, expected %d", Dd, D);
    ALIVE_CHECK_ARG(M >= 1 && M < (int64_t)1 << 31, "alive_library_pack: M out of range");
    int64_t M_pad = alive_library_padded_rows(M);
    lib_pack_kernel<<<(unsigned)(M_pad / 64), 256, 0, (hipStream_t)stream>>>(tokens, M, M_pad, (unsigned short*)lib_bf16,
                                                                            rows_f32, norms);
    ALIVE_CHECK_LAUNCH("alive_library_pack");
    return ALIVE_OK;
}

static int knn_scan_launch(const float* src, int T, int64_t Tt, const float* rows_f32, const float* norms, int64_t M,
                           int64_t idx_base, int k, float* s_f32, unsigned short* s_bf16, float* pv, int* pi, float* out_val,
                           int32_t* out_idx, int* stats, hipStream_t s, hipEvent_t g_ev_start, hipEvent_t g_ev_stop) {
    stats_init_kernel<<<1, 64, 0, s>>>(stats, TIER_SCAN);
    src_prep_small_kernel<<<(unsigned)Tt, 256, 0, s>>>(src, T, Tt, s_f32, s_bf16, nullptr);
    int blocks = (int)((M + 4 * SCAN_WAVES - 1) / (4 * SCAN_WAVES));          // >= 4 rows per wave
    if (blocks > SCAN_MAX_LISTS / SCAN_WAVES) blocks = SCAN_MAX_LISTS / SCAN_WAVES;
    if (blocks < 1) blocks = 1;
    if (g_ev_start) (void)hipEventRecord(g_ev_start, s);
    knn_scan_kernel<true><<<blocks, 64 * SCAN_WAVES, 0, s>>>(s_f32, rows_f32, norms, M, (int)Tt, k, pv, pi);
    if (g_ev_stop) (void)hipEventRecord(g_ev_stop, s);
    if (k <= 4) knn_scan_merge_kernel<true><<<(unsigned)Tt, 256, 0, s>>>(pv, pi, blocks * SCAN_WAVES, k, idx_base, out_val, out_idx);
    else knn_scan_merge_kernel<false><<<(unsigned)Tt, 256, 0, s>>>(pv, pi, blocks * SCAN_WAVES, k, idx_base, out_val, out_idx);
    ALIVE_CHECK_LAUNCH("alive_knn_search(scan)");
    return ALIVE_OK;
}

// exact tier over a frame list with a device-side count (list == nullptr: all Tt frames)
static void knn_exact_launch(const SearchWs& w, const float* rows_f32, const float* norms, int64_t M, int64_t Tt, int64_t idx_base,
                             int k, const int* list, const int* cnt, float* out_val, int32_t* out_idx, hipStream_t s,
                             int max_count = 0x7fffffff) {
#define ALIVE_EXACT(G_)                                                                                                        \
    {                                                                                                                          \
        knn_exact_kernel<G_><<<EX_BLOCKS, 256, 0, s>>>(w.s_f32, rows_f32, norms, M, Tt, k, list, cnt, w.pv, w.pi, max_count);   \
        knn_exact_merge_kernel<G_><<<1024, 256, 0, s>>>(w.pv, w.pi, Tt, k, idx_base, list, cnt, out_val, out_idx, max_count);  \
    }
    switch (exact_group(k)) {
        case 16: ALIVE_EXACT(16) break;
        case 8: ALIVE_EXACT(8) break;
        case 4: ALIVE_EXACT(4) break;
        case 2: ALIVE_EXACT(2) break;
        default: ALIVE_EXACT(1) break;
    }
#undef ALIVE_EXACT
}

static void src_prep_launch(const SearchWs& w, const float* src, int T, int64_t Tt, hipStream_t s) {
    float* dq = w.det_q != nullptr ? w.dq : nullptr;
    if (Tt <= 512) src_prep_small_kernel<<<(unsigned)w.p16.Tt_pad, 256, 0, s>>>(src, T, Tt, w.s_f32, w.s_bf16, dq);
    else src_prep_kernel<<<(unsigned)(w.p16.Tt_pad / 64), 256, 0, s>>>(src, T, Tt, w.p16.Tt_pad, w.s_f32, w.s_bf16, dq);
}

static int check_search_args(const char* what, const void* a, const void* b, int N, int T, int k, int64_t M) {
    ALIVE_CHECK_ARG(a && b, "%s: null pointer", what);
    ALIVE_CHECK_ARG(N > 0 && T > 0, "%s: empty source", what);
    ALIVE_CHECK_ARG(k >= 1 && k <= ALIVE_MAX_K, "%s: k=%d outside [1,%d]", what, k, ALIVE_MAX_K);
    ALIVE_CHECK_ARG(M >= k, "%s: library shard has %lld vectors, fewer than k=%d", what, (long long)M, k);
    ALIVE_CHECK_ARG((int64_t)N * T < ((int64_t)1 << 31) - FT6, "%s: %lld frames in one call", what, (long long)N * T);
    return ALIVE_OK;
}

static int lds_optin(const char* what) {
    static LdsOptIn optin8, optin6, optin16, optin_split;
    hipError_t e = optin8.ensure({(const void*)knn_score8_kernel, (const void*)knn_probe8_kernel}, SCORE8_LDS);
    if (e == hipSuccess) e = optin6.ensure({(const void*)knn_score6_kernel, (const void*)knn_probe6_kernel}, SCORE6_LDS);
    if (e == hipSuccess) e = optin16.ensure({(const void*)knn_score_kernel<false>, (const void*)knn_score_kernel<true>}, SCORE_LDS);
    if (e == hipSuccess) e = optin_split.ensure({(const void*)knn_collect_split_kernel}, SPLIT_LDS);
    if (e != hipSuccess) {
        alive_set_error("%s: cannot reserve %d B of LDS: %s", what, SCORE_LDS, hipGetErrorString(e));
        return ALIVE_ERR_LAUNCH;
    }
    return ALIVE_OK;
}

extern "C" size_t alive_library_fp8_bytes(int64_t M) { return (size_t)alive_library_padded_rows(M) * D; }

extern "C" int alive_library_pack_fp8(const void* lib_bf16, int64_t M, void* lib_f8, void* stream) {
    ALIVE_CHECK_ARG(lib_bf16 && lib_f8 && M >= 1, "alive_library_pack_fp8: bad args");
    const int64_t n8 = alive_library_padded_rows(M) * D / 8;
    lib_to_fp8_kernel<<<(unsigned)((n8 + 255) / 256), 256, 0, (hipStream_t)stream>>>((const unsigned short*)lib_bf16, n8, (uint2*)lib_f8);
    ALIVE_CHECK_LAUNCH("alive_library_pack_fp8");
    return ALIVE_OK;
}

// fp6 e2m3 form of the normalised rows (x 2^5), alive_library_fp8_bytes(M) bytes like the fp8 form: every group of 32 features keeps
// a 32-byte slot (24 bytes of codes + 8 of zeros), so that the tile image, the LDS-DMA pieces and the fragment reads are the fp8 kernel's
extern "C" int alive_library_pack_fp6(const void* lib_bf16, int64_t M, void* lib_f6, void* stream) {
    ALIVE_CHECK_ARG(lib_bf16 && lib_f6 && M >= 1, "alive_library_pack_fp6: bad args");
    const int64_t n32 = alive_library_padded_rows(M) * D / 32;
    to_fp6_kernel<<<(unsigned)((n32 + 255) / 256), 256, 0, (hipStream_t)stream>>>((const unsigned short*)lib_bf16, n32, n32, (u32x4*)lib_f6);
    ALIVE_CHECK_LAUNCH("alive_library_pack_fp6");
    return ALIVE_OK;
}

// alive_knn_search_strict with a lo-plane library (lib_lo != NULL) additionally needs both bf16 planes of the frames (they come last in
// the layout).  alive_knn_workspace_bytes is the bound that is safe for EVERY search entry point (a caller built against an older header
// sizes one buffer with it and may hand it to the strict search, which takes no size argument); the searches without a lo plane need
// only alive_knn_workspace_bytes_fast (3 KB per frame less).
extern "C" size_t alive_knn_workspace_bytes_fast(int64_t Tt, int64_t M) {
    // sized for any k the entry points accept (the exact tier's partial lists grow with ceil(Tt / G(k)))
    const size_t a = ws_layout(nullptr, Tt, M, 4).bytes, b = ws_layout(nullptr, Tt, M, ALIVE_MAX_K).bytes;
    return a > b ? a : b;
}
extern "C" size_t alive_knn_workspace_bytes_strict(int64_t Tt, int64_t M) {
    const size_t a = ws_layout(nullptr, Tt, M, 4, true).bytes, b = ws_layout(nullptr, Tt, M, ALIVE_MAX_K, true).bytes;
    return a > b ? a : b;
}
extern "C" size_t alive_knn_workspace_bytes(int64_t Tt, int64_t M) { return alive_knn_workspace_bytes_strict(Tt, M); }

// The collect tier: the frames whose bf16 certificate failed (list1, with the thresholds thr1 the rescoring kernel recorded
// for them) go through the bf16 scoring kernel once more, in its COLLECT form -- every row whose stage score reaches the
// frame's threshold is kept and rescored exactly, together with the frame's current top-k.  A row below the threshold
// cannot belong to the top-k (by the same slack the certificate was tested with: 7 sigma of the measured stage error, or
// the deterministic bound of the strict search), so the result is final; frames with more rows above the threshold than
// the lists hold (dense clusters: more rows inside the stage's error than any candidate list can keep) are the only ones
// left for the exact scan.  Same two launch plans as the re-search: few frames (<= fcap) / the whole batch.
static void collect_tier_launch(const SearchWs& w, const void* lib_bf16, const float* rows_f32, const float* norms, int64_t M,
                                int64_t Tt, int64_t idx_base, int k, float* out_val, int32_t* out_idx, hipStream_t s) {
    int* cnt1 = w.stats + ST_FLAG16;
    int* cnt2 = w.stats + ST_FLAGC;
    const int fcap = w.fcap;
    if (w.lib_lo != nullptr) {
        // Strict search with a lo-plane library: the frames that failed the deterministic certificate (list1, thresholds
        // thr1 = v_k - SPLIT_BOUND) go STRAIGHT to the split-bf16 collect pass -- on a dense library 95 % of them overflow the
        // single-plane collect tier (its band is the certificate's 1.8e-3), which would be a wasted pass over the library -- its
        // rows are rescored exactly, and only what overflows it (clusters of near-copies) reaches the exact scan.
        const unsigned Tp = (unsigned)w.p16.Tt_pad;
        knn_exact_launch(w, rows_f32, norms, M, Tt, idx_base, k, w.list1, cnt1, out_val, out_idx, s, COLLECT_MIN);
        gather_frames_split_kernel<<<Tp, 128, 0, s>>>(w.s_bf16, w.s_f32, w.list1, cnt1, COLLECT_MIN, w.s_c2h, w.s_c2l);
        const int caps = w.rows_b / w.p16.P < 64 ? w.rows_b / w.p16.P : 64;              // P <= MAX_SPLIT = 64: at least 4
        knn_collect_split_kernel<<<dim3(Tp / FT2, w.p16.split), 256, SPLIT_LDS, s>>>(
            w.s_c2h, w.s_c2l, (const unsigned short*)lib_bf16, w.lib_lo, M, w.p16.tiles_total, w.p16.tiles_per_split, w.p16.P, caps,
            w.c2v, w.c2i, cnt1, COLLECT_MIN, w.thr1);
        rescore_launch((unsigned)((Tt + 3) / 4), s, w.c2v, w.c2i, w.p16.P, caps, w.s_f32, rows_f32, norms, Tt, idx_base, k,
                                                                   out_val, out_idx, w.list1, cnt1, COLLECT_MIN, 0x7fffffff, w.list2, cnt2,
                                                                   CERT_Z, KH, 1.0f, SD_PRIOR16, nullptr, nullptr, nullptr, 1, 0.0f);
        knn_exact_launch(w, rows_f32, norms, M, Tt, idx_base, k, w.list2, cnt2, out_val, out_idx, s);
        return;
    }
    // a handful of frames: the exact scan costs less than the fixed cost of one more scoring pass over the library (~8 ms at 1 M
    // rows against ~20 us per frame of exact scan)
    const int caps_t = w.rows_t / w.pt.P < 64 ? w.rows_t / w.pt.P : 64, caps_b = w.rows_b / w.p16.P < 64 ? w.rows_b / w.p16.P : 64;
    knn_exact_launch(w, rows_f32, norms, M, Tt, idx_base, k, w.list1, cnt1, out_val, out_idx, s, COLLECT_MIN);
    gather_frames_kernel<<<(unsigned)fcap, 128, 0, s>>>(w.s_bf16, w.list1, cnt1, COLLECT_MIN, fcap, w.s_c);
    knn_score_kernel<true><<<dim3((unsigned)(fcap / FT), w.pt.split), 256, SCORE_LDS, s>>>(
        w.s_c, (const unsigned short*)lib_bf16, M, w.pt.tiles_total, w.pt.tiles_per_split, w.pt.P, w.c2v, w.c2i, cnt1, COLLECT_MIN, fcap, 1,
        w.thr1, SeedArgs{nullptr, nullptr, 0, 0.0f, nullptr}, caps_t);
    rescore_launch((unsigned)((fcap + 3) / 4), s, w.c2v, w.c2i, w.pt.P, caps_t, w.s_f32, rows_f32, norms, fcap, idx_base, k,
                                                                 out_val, out_idx, w.list1, cnt1, COLLECT_MIN, fcap, w.list2, cnt2, CERT_Z,
                                                                 KH, 1.0f, SD_PRIOR16, nullptr, nullptr, nullptr, 1, 0.0f);
    if (w.p16.Tt_pad > fcap) {
        gather_frames_kernel<<<(unsigned)w.p16.Tt_pad, 128, 0, s>>>(w.s_bf16, w.list1, cnt1, fcap, 0x7fffffff, w.s_c);
        knn_score_kernel<true><<<dim3((unsigned)(w.p16.Tt_pad / FT), w.p16.split), 256, SCORE_LDS, s>>>(
            w.s_c, (const unsigned short*)lib_bf16, M, w.p16.tiles_total, w.p16.tiles_per_split, w.p16.P, w.c2v, w.c2i, cnt1, fcap,
            0x7fffffff, 1, w.thr1, SeedArgs{nullptr, nullptr, 0, 0.0f, nullptr}, caps_b);
        rescore_launch((unsigned)((Tt + 3) / 4), s, w.c2v, w.c2i, w.p16.P, caps_b, w.s_f32, rows_f32, norms, Tt, idx_base, k,
                                                                   out_val, out_idx, w.list1, cnt1, fcap, 0x7fffffff, w.list2, cnt2,
                                                                   CERT_Z, KH, 1.0f, SD_PRIOR16, nullptr, nullptr, nullptr, 1, 0.0f);
    }
    knn_exact_launch(w, rows_f32, norms, M, Tt, idx_base, k, w.list2, cnt2, out_val, out_idx, s);
}

// The bf16 re-search of the frames in list[0 .. *cnt) (compacted), certified, behind either first stage:
//   up to RESEARCH_MIN frames: straight to the exact scan;
//   tier 1 (.. fcap frames): library split chosen for few frames;
//   tier 2 (more): the split of the whole batch; blocks past the count exit at once.
// Frames that fail the bf16 certificate land in list1 and go through the exact scan.
static void bf16_tiers_launch(const SearchWs& w, const void* lib_bf16, const float* rows_f32, const float* norms, int64_t M,
                              int64_t Tt, int64_t idx_base, int k, float* out_val, int32_t* out_idx, hipStream_t s) {
    int* cnt0 = w.stats + ST_FLAG8;
    int* cnt1 = w.stats + ST_FLAG16;
    const int fcap = w.fcap;
    knn_exact_launch(w, rows_f32, norms, M, Tt, idx_base, k, w.list0, cnt0, out_val, out_idx, s, RESEARCH_MIN);
    // tier 1a (round 5): RESEARCH_MIN < count <= one frame block.  Tier 1's split is chosen for 4096 frames (16 frame blocks x 16
    // splits): a hundred flagged frames keep ONE of its frame blocks busy, i.e. 16 CUs walk the whole library -- 9 ms at 1 M rows,
    // which is what the 73 frames the fp6 stage leaves uncertified on the bench batch cost.  Here the one block's library is cut
    // MAX_SPLIT ways (64 CUs, 1024 candidates per frame = what the rescoring kernel takes).  Same buffers as tier 1: the gates
    // (RESEARCH_MIN, TIER1A] / (TIER1A, fcap] exclude each other.
    const int t1a = fcap < TIER1A ? fcap : TIER1A;
    gather_frames_kernel<<<(unsigned)t1a, 128, 0, s>>>(w.s_bf16, w.list0, cnt0, RESEARCH_MIN, t1a, w.s_c);
    knn_score_kernel<false><<<dim3(1, w.pa.split), 256, SCORE_LDS, s>>>(
        w.s_c, (const unsigned short*)lib_bf16, M, w.pa.tiles_total, w.pa.tiles_per_split, w.pa.P, w.cv1, w.ci1, cnt0, RESEARCH_MIN, t1a, 1, nullptr, SeedArgs{nullptr, nullptr, 0, 0.0f, nullptr}, 0);
    rescore_launch((unsigned)((t1a + 3) / 4), s, w.cv1, w.ci1, w.pa.P, KP, w.s_f32, rows_f32, norms, t1a, idx_base, k,
                                                                out_val, out_idx, w.list0, cnt0, RESEARCH_MIN, t1a, w.list1, cnt1, CERT_Z, KH, 1.0f, SD_PRIOR16, w.det_q, w.det_lib, w.thr1, 0, w.lib_lo != nullptr ? SPLIT_BOUND : 0.0f);
    if (fcap > t1a) {
    gather_frames_kernel<<<(unsigned)fcap, 128, 0, s>>>(w.s_bf16, w.list0, cnt0, t1a, fcap, w.s_c);
    knn_score_kernel<false><<<dim3((unsigned)(fcap / FT), w.pt.split), 256, SCORE_LDS, s>>>(
        w.s_c, (const unsigned short*)lib_bf16, M, w.pt.tiles_total, w.pt.tiles_per_split, w.pt.P, w.cv1, w.ci1, cnt0, t1a, fcap, 1, nullptr, SeedArgs{nullptr, nullptr, 0, 0.0f, nullptr}, 0);
    rescore_launch((unsigned)((fcap + 3) / 4), s, w.cv1, w.ci1, w.pt.P, KP, w.s_f32, rows_f32, norms, fcap, idx_base, k,
                                                                 out_val, out_idx, w.list0, cnt0, t1a, fcap, w.list1, cnt1, CERT_Z, KH, 1.0f, SD_PRIOR16, w.det_q, w.det_lib, w.thr1, 0, w.lib_lo != nullptr ? SPLIT_BOUND : 0.0f);
    }
    if (w.p16.Tt_pad > fcap) {
        gather_frames_kernel<<<(unsigned)w.p16.Tt_pad, 128, 0, s>>>(w.s_bf16, w.list0, cnt0, fcap, 0x7fffffff, w.s_c);
        // (the frames are compacted: a block's 256 slots are the same in every split, so the seeds are indexed by slot)
        const SeedArgs sa = seeds_for(w, w.p16.Tt_pad / FT, w.p16.split, k, w.det_q != nullptr ? SEED_MARGIN16_STRICT : SEED_MARGIN16,
                                      ST_SEEDED16, s);
        knn_score_kernel<false><<<dim3((unsigned)(w.p16.Tt_pad / FT), w.p16.split), 256, SCORE_LDS, s>>>(
            w.s_c, (const unsigned short*)lib_bf16, M, w.p16.tiles_total, w.p16.tiles_per_split, w.p16.P, w.cv, w.ci, cnt0, fcap,
            0x7fffffff, 1, nullptr, sa, 0);
        rescore_launch((unsigned)((Tt + 3) / 4), s, w.cv, w.ci, w.p16.P, KP, w.s_f32, rows_f32, norms, Tt, idx_base, k,
                                                                   out_val, out_idx, w.list0, cnt0, fcap, 0x7fffffff, w.list1, cnt1,
                                                                   CERT_Z, KH, 1.0f, SD_PRIOR16, w.det_q, w.det_lib, w.thr1, 0, w.lib_lo != nullptr ? SPLIT_BOUND : 0.0f);
    }
    collect_tier_launch(w, lib_bf16, rows_f32, norms, M, Tt, idx_base, k, out_val, out_idx, s);
}

// Search with the bf16 MFMA as the first candidate stage: every frame's candidate set is certified against the bf16 score
// error measured on its own rescored candidates; frames that fail go through the exact fp32 scan.  k > 8 (deeper than a
// half-list of the candidate stage): the exact scan for every frame.
static int knn_search_impl(const float* src, int N, int T, const void* lib_bf16, const float* rows_f32,
                           const float* norms, int64_t M, int64_t idx_base, int k, float* out_val, int32_t* out_idx,
                           void* ws, void* stream, hipEvent_t g_ev_start, hipEvent_t g_ev_stop, const float* strict_bound = nullptr,
                           const void* lib_lo = nullptr) {
    ALIVE_CHECK_ARG(src && lib_bf16 && rows_f32 && norms && out_val && out_idx && ws, "alive_knn_search: null pointer");
    if (int rc = check_search_args("alive_knn_search", src, ws, N, T, k, M)) return rc;
    const int64_t Tt = (int64_t)N * T;
    SearchWs w = ws_layout(ws, Tt, M, k, strict_bound != nullptr && lib_lo != nullptr);
    if (strict_bound != nullptr) {         // deterministic certificate: per-frame rounding error + the library's (alive_library_rounding_bound)
        w.det_q = w.dq;
        w.det_lib = strict_bound;
        w.lib_lo = (const unsigned short*)lib_lo;      // nullptr: no split tier, collect-tier overflow goes straight to the exact scan
    }
    hipStream_t s = (hipStream_t)stream;
    if (Tt * k <= 64 && M <= SCAN_ROWS_MAX)            // a handful of frames: exact fp32 scan of the rows, no candidate stage
        return knn_scan_launch(src, T, Tt, rows_f32, norms, M, idx_base, k, w.s_f32, w.s_bf16, w.pv, w.pi, out_val, out_idx, w.stats, s, g_ev_start, g_ev_stop);
    if (int rc = lds_optin("alive_knn_search")) return rc;
    stats_init_kernel<<<1, 64, 0, s>>>(w.stats, k > KH ? TIER_EXACT_ALL : TIER_BF16);
    src_prep_launch(w, src, T, Tt, s);
    if (k > KH) {
        if (g_ev_start) (void)hipEventRecord(g_ev_start, s);
        knn_exact_launch(w, rows_f32, norms, M, Tt, idx_base, k, nullptr, nullptr, out_val, out_idx, s);
        if (g_ev_stop) (void)hipEventRecord(g_ev_stop, s);
        ALIVE_CHECK_LAUNCH("alive_knn_search(exact)");
        return ALIVE_OK;
    }
    const SearchPlan& p = w.p16;
    const SeedArgs sa = seeds_for(w, p.Tt_pad / FT, p.split, k, w.det_q != nullptr ? SEED_MARGIN16_STRICT : SEED_MARGIN16, ST_SEEDED16, s);
    if (g_ev_start) (void)hipEventRecord(g_ev_start, s);             // behind the memset of the seed flags: the events bracket the kernel alone
    knn_score_kernel<false><<<dim3((unsigned)(p.Tt_pad / FT), p.split), 256, SCORE_LDS, s>>>(
        w.s_bf16, (const unsigned short*)lib_bf16, M, p.tiles_total, p.tiles_per_split, p.P, w.cv, w.ci, nullptr, 0, 0, 0, nullptr, sa, 0);
    if (g_ev_stop) (void)hipEventRecord(g_ev_stop, s);
    rescore_launch((unsigned)((Tt + 3) / 4), s, w.cv, w.ci, p.P, KP, w.s_f32, rows_f32, norms, Tt, idx_base, k,
                                                               out_val, out_idx, nullptr, nullptr, 0, 0, w.list1, w.stats + ST_FLAG16,
                                                               CERT_Z, KH, 1.0f, SD_PRIOR16, w.det_q, w.det_lib, w.thr1, 0, w.lib_lo != nullptr ? SPLIT_BOUND : 0.0f);
    collect_tier_launch(w, lib_bf16, rows_f32, norms, M, Tt, idx_base, k, out_val, out_idx, s);
    ALIVE_CHECK_LAUNCH("alive_knn_search");
    return ALIVE_OK;
}

// The same search with the candidate stage on the fp8 MFMA (knn_score8_kernel); lib_f8 from alive_library_pack_fp8.
//   probe   (batches of >= 16384 frames) the fp8 stage + its certificate on a sample of 1024 frames; when more than 40 % of
//           them fail -- a library whose best cosines lie closer together than the fp8 error -- the fp8 pass over the
//           batch is skipped (mode 1) and every frame goes straight to the bf16 stage;
//   fp8     (mode 0) candidate stage on the fp8 MFMA, exact rescoring, certificate -> list0;
//   bf16    the frames of list0 (all of them in mode 1) through the bf16 candidate stage, exact rescoring, certificate
//           with the bf16 error statistics -> list1;
//   exact   the frames of list1 through the brute-force fp32 scan.
// All of it is launched up front -- the kernels read the counters on the device and return at once when a tier is empty.
extern "C" int alive_knn_search(const float* src, int N, int T, const void* lib_bf16, const float* rows_f32,
                                const float* norms, int64_t M, int64_t idx_base, int k, float* out_val, int32_t* out_idx,
                                void* ws, void* stream) {
    return knn_search_impl(src, N, T, lib_bf16, rows_f32, norms, M, idx_base, k, out_val, out_idx, ws, stream, nullptr, nullptr);
}
// Strict search: the bf16 candidate stage with a DETERMINISTIC certificate (knn_rescore_kernel: det_bound), frames that
// fail it through the exact fp32 scan.  No statistical assumption anywhere: the result is the exact top-k of the rescoring
// arithmetic for every input.  bound: device float[1] from alive_library_rounding_bound.
extern "C" int alive_knn_search_strict(const float* src, int N, int T, const void* lib_bf16, const void* lib_lo, const float* rows_f32,
                                       const float* norms, const float* bound, int64_t M, int64_t idx_base, int k, float* out_val,
                                       int32_t* out_idx, void* ws, void* stream, void* ev_start, void* ev_stop) {
    ALIVE_CHECK_ARG(bound != nullptr, "alive_knn_search_strict: null bound (alive_library_rounding_bound)");
    return knn_search_impl(src, N, T, lib_bf16, rows_f32, norms, M, idx_base, k, out_val, out_idx, ws, stream,
                           (hipEvent_t)ev_start, (hipEvent_t)ev_stop, bound, lib_lo);
}

// lo plane of a packed library (strict searches): lib_lo[M_pad][768] = bf16(r^ - lib_bf16), 2 * 768 * alive_library_padded_rows(M) bytes
extern "C" int alive_library_pack_lo(const void* lib_bf16, const float* rows_f32, const float* norms, int64_t M, void* lib_lo,
                                     void* stream) {
    ALIVE_CHECK_ARG(lib_bf16 && rows_f32 && norms && lib_lo && M >= 1, "alive_library_pack_lo: bad args");
    const int64_t M_pad = alive_library_padded_rows(M);
    lib_lo_kernel<<<(unsigned)((M_pad * D + 255) / 256), 256, 0, (hipStream_t)stream>>>((const unsigned short*)lib_bf16, rows_f32, norms, M,
                                                                                     M_pad, (unsigned short*)lib_lo);
    ALIVE_CHECK_LAUNCH("alive_library_pack_lo");
    return ALIVE_OK;
}

extern "C" int alive_library_rounding_bound(const void* lib_bf16, const float* rows_f32, const float* norms, int64_t M,
                                            float* bound, void* stream) {
    ALIVE_CHECK_ARG(lib_bf16 && rows_f32 && norms && bound && M >= 1, "alive_library_rounding_bound: bad arguments");
    (void)hipMemsetAsync(bound, 0, sizeof(float), (hipStream_t)stream);
    lib_rounding_bound_kernel<<<(unsigned)((M + 3) / 4), 256, 0, (hipStream_t)stream>>>((const unsigned short*)lib_bf16, rows_f32, norms,
                                                                                        M, (unsigned*)bound);
    ALIVE_CHECK_LAUNCH("alive_library_rounding_bound");
    return ALIVE_OK;
}

extern "C" int alive_knn_search_timed(const float* src, int N, int T, const void* lib_bf16, const float* rows_f32,
                                      const float* norms, int64_t M, int64_t idx_base, int k, float* out_val, int32_t* out_idx,
                                      void* ws, void* stream, void* ev_start, void* ev_stop) {
    return knn_search_impl(src, N, T, lib_bf16, rows_f32, norms, M, idx_base, k, out_val, out_idx, ws, stream,
                           (hipEvent_t)ev_start, (hipEvent_t)ev_stop);
}

static int knn_search_fp8_impl(const float* src, int N, int T, const void* lib_f8, const void* lib_bf16, const float* rows_f32,
                               const float* norms, int64_t M, int64_t idx_base, int k, float* out_val, int32_t* out_idx,
                               void* ws, void* stream, hipEvent_t g_ev_start, hipEvent_t g_ev_stop, int fmt = 0,
                               const float* y_rot = nullptr, float rot_c0 = 0.0f) {
    // y_rot != NULL (fp8 only): lib_f8 holds the ROTATED codes of the rows (alive_library_pack_fp8_rot) and y_rot the frames' rotated
    // coordinates [N][576][T]: the frames' codes come from rot_codes_kernel, the stage's error prior is SD_PRIOR8R; everything else --
    // exact rescoring on the original rows, certificates, the bf16 tiers on lib_bf16 -- is the plain search's
    ALIVE_CHECK_ARG(src && lib_f8 && lib_bf16 && rows_f32 && norms && out_val && out_idx && ws, "alive_knn_search_fp8: null pointer");
    if (int rc = check_search_args("alive_knn_search_fp8", src, ws, N, T, k, M)) return rc;
    const int64_t Tt = (int64_t)N * T;
    if (k > KH)
        return knn_search_impl(src, N, T, lib_bf16, rows_f32, norms, M, idx_base, k, out_val, out_idx, ws, stream, g_ev_start, g_ev_stop);
    const SearchWs w = ws_layout(ws, Tt, M, k);
    const bool f6 = fmt == 2;
    const SearchPlan& p = f6 ? w.p6 : w.p8;
    const SearchPlan& pp = f6 ? w.pp6 : w.pp;
    const int ft = f6 ? FT6 : FT, probe_pad = f6 ? w.probe_pad6 : w.probe_pad, lds = f6 ? SCORE6_LDS : SCORE8_LDS;
    const float pre = f6 ? 1.0f / (F6_SCALE * F6_SCALE) : 1.0f / (F8_SCALE * F8_SCALE),
                prior = f6 ? SD_PRIOR6 : (y_rot != nullptr ? SD_PRIOR8R : SD_PRIOR8);
    hipStream_t s = (hipStream_t)stream;
    if (Tt * k <= 64 && M <= SCAN_ROWS_MAX)            // streaming ring: the exact scan, no candidate stage at all
        return knn_scan_launch(src, T, Tt, rows_f32, norms, M, idx_base, k, w.s_f32, w.s_bf16, w.pv, w.pi, out_val, out_idx, w.stats, s, g_ev_start, g_ev_stop);
    if (int rc = lds_optin("alive_knn_search_fp8")) return rc;
    // (history tag: library size, frame count and stage -- a workspace that was last used for another search starts over)
    const int hist_tag = (int)(((uint64_t)M * 0x9E3779B1u) ^ ((uint64_t)Tt * 0x85EBCA77u) ^ (f6 ? 0x6u : (y_rot != nullptr ? 0x18u : 0x8u))) | 1;
    stats_init_kernel<<<1, 64, 0, s>>>(w.stats, f6 ? TIER_FP6 : TIER_FP8, w.probe_n > 0 ? hist_tag : 0);
    src_prep_launch(w, src, T, Tt, s);
    if (f6) {
        const int64_t n32 = p.Tt_pad * D / 32;
        (void)hipMemsetAsync(w.clip6, 0, (size_t)p.Tt_pad, s);
        to_fp6_kernel<<<(unsigned)((n32 + 255) / 256), 256, 0, s>>>(w.s_bf16, n32, w.p16.Tt_pad * D / 32, (u32x4*)w.s_f8, w.clip6);
    } else if (y_rot != nullptr) {
        rot_codes_kernel<<<(unsigned)(p.Tt_pad / 32), 256, 0, s>>>(y_rot, Tt, p.Tt_pad, T, 1, rot_c0, w.s_f8);
    } else {
        const int64_t n8 = p.Tt_pad * D / 8;
        src_to_fp8_kernel<<<(unsigned)((n8 + 255) / 256), 256, 0, s>>>(w.s_bf16, n8, (uint2*)w.s_f8);
    }
    int* mode = w.stats + ST_MODE;
    if (w.probe_n > 0) {
        const int* pgate = w.stats + ST_PROBE_GATE;      // 0: the history of this workspace says no probe is needed (all five launches return at once)
        probe_gather_kernel<<<(unsigned)probe_pad, 64, 0, s>>>(w.s_f8, Tt, w.probe_n, probe_pad, w.s_p8, w.p_list, pgate);
        if (f6)
            knn_probe6_kernel<<<dim3((unsigned)(probe_pad / ft), pp.split), 256, lds, s>>>(
                w.s_p8, (const unsigned char*)lib_f8, M, pp.tiles_total, pp.tiles_per_split, pp.P, w.cvp, w.cip, pgate);
        else
            knn_probe8_kernel<<<dim3((unsigned)(probe_pad / ft), pp.split), 256, lds, s>>>(
                w.s_p8, (const unsigned char*)lib_f8, M, pp.tiles_total, pp.tiles_per_split, pp.P, w.cvp, w.cip, pgate);
        // the sample's own rescoring.  The kernel writes a frame's result to the frame's own output rows (out[frame]), so
        // the sample's exact lists land in the caller's outputs and are overwritten by the pass over the batch; its flagged
        // frames (second half of p_list) are only counted
        rescore_launch((unsigned)((w.probe_n + 3) / 4), s, w.cvp, w.cip, pp.P, KP8, w.s_f32, rows_f32, norms, w.probe_n,
                                                                          idx_base, k, out_val, out_idx, w.p_list, pgate, GATE_BOOL, 0,
                                                                          w.p_list + probe_pad, w.stats + ST_PROBE_CNT, CERT_Z, KH8,
                                                                          pre, prior, nullptr, nullptr, nullptr, 0, 0.0f, f6 ? w.clip6 : nullptr);
        probe_decide_kernel<<<1, 1, 0, s>>>(w.stats, w.probe_n, PROBE_NUM, PROBE_DEN);
    }
    // ---- mode 0: the fp8 / fp6 stage first ----
    const SeedArgs sa8 = seeds_for(w, p.Tt_pad / ft, p.split, k, f6 ? seed_margin6() : seed_margin8(), ST_SEEDED, s, f6 ? SEED_MIN_FB6 : SEED_MIN_FB);
    if (g_ev_start) (void)hipEventRecord(g_ev_start, s);             // behind the memset of the seed flags: the events bracket the kernel alone
    if (f6)
        knn_score6_kernel<<<dim3((unsigned)(p.Tt_pad / ft), p.split), 256, lds, s>>>(
            w.s_f8, (const unsigned char*)lib_f8, M, p.tiles_total, p.tiles_per_split, p.P, w.cv, w.ci, mode, -1, 0, sa8);
    else
        knn_score8_kernel<<<dim3((unsigned)(p.Tt_pad / ft), p.split), 256, lds, s>>>(
            w.s_f8, (const unsigned char*)lib_f8, M, p.tiles_total, p.tiles_per_split, p.P, w.cv, w.ci, mode, -1, 0, sa8);
    if (g_ev_stop) (void)hipEventRecord(g_ev_stop, s);
    rescore_launch((unsigned)((Tt + 3) / 4), s, w.cv, w.ci, p.P, KP8, w.s_f32, rows_f32, norms, Tt, idx_base, k,
                                                               out_val, out_idx, nullptr, mode, -1, 0, w.list0, w.stats + ST_FLAG8,
                                                               CERT_Z, KH8, pre, prior, nullptr, nullptr, nullptr, 0, 0.0f, f6 ? w.clip6 : nullptr);
    // ---- mode 1: bf16 first (every frame into list0) ----
    if (w.probe_n > 0) flag_all_kernel<<<(unsigned)((Tt + 255) / 256), 256, 0, s>>>(w.list0, w.stats + ST_FLAG8, Tt, mode, 0, 1);
    if (w.probe_n > 0) probe_history_kernel<<<1, 1, 0, s>>>(w.stats, Tt);        // (ST_FLAG8 is final here: the tiers below only read it)
    bf16_tiers_launch(w, lib_bf16, rows_f32, norms, M, Tt, idx_base, k, out_val, out_idx, s);
    ALIVE_CHECK_LAUNCH("alive_knn_search_fp8");
    return ALIVE_OK;
}

extern "C" int alive_knn_search_fp8(const float* src, int N, int T, const void* lib_f8, const void* lib_bf16, const float* rows_f32,
                                    const float* norms, int64_t M, int64_t idx_base, int k, float* out_val, int32_t* out_idx,
                                    void* ws, void* stream) {
    return knn_search_fp8_impl(src, N, T, lib_f8, lib_bf16, rows_f32, norms, M, idx_base, k, out_val, out_idx, ws, stream, nullptr, nullptr);
}
extern "C" int alive_knn_search_fp8_timed(const float* src, int N, int T, const void* lib_f8, const void* lib_bf16, const float* rows_f32,
                                          const float* norms, int64_t M, int64_t idx_base, int k, float* out_val, int32_t* out_idx,
                                          void* ws, void* stream, void* ev_start, void* ev_stop) {
    return knn_search_fp8_impl(src, N, T, lib_f8, lib_bf16, rows_f32, norms, M, idx_base, k, out_val, out_idx, ws, stream,
                               (hipEvent_t)ev_start, (hipEvent_t)ev_stop);
}

// The fp8 search on ROTATED operands (dense banks; rot_codes_kernel above).  y_rows: [count][576] fp32 coordinates of the unit rows
// m0 .. m0 + count - 1 in the bank's basis (the caller may pack a big bank in chunks); lib_f8: alive_library_fp8_bytes(M) bytes.
extern "C" int alive_knn_rot_coordinates(void) { return ROT_C; }
extern "C" int alive_knn_rot_leading(void) { return ROT_A; }
extern "C" int alive_knn_rot_mixed(void) { return ROT_RHO; }
extern "C" int alive_library_pack_fp8_rot(const float* y_rows, int64_t m0, int64_t count, int64_t M, float c0, void* lib_f8, void* stream) {
    ALIVE_CHECK_ARG(y_rows && lib_f8 && M >= 1 && m0 >= 0 && count >= 1 && m0 + count <= M && (m0 % 32) == 0,
                    "alive_library_pack_fp8_rot: bad args (m0 must be a multiple of 32)");
    const int64_t m_pad = alive_library_padded_rows(M);
    const int64_t span = m0 + count == M ? m_pad - m0 : count;          // the last chunk also zeroes the padding rows
    ALIVE_CHECK_ARG(span % 32 == 0, "alive_library_pack_fp8_rot: a chunk that is not the last must hold a multiple of 32 rows");
    rot_codes_kernel<<<(unsigned)(span / 32), 256, 0, (hipStream_t)stream>>>(y_rows, count, span, 0, 0, c0, (unsigned char*)lib_f8 + (size_t)m0 * D);
    ALIVE_CHECK_LAUNCH("alive_library_pack_fp8_rot");
    return ALIVE_OK;
}
extern "C" int alive_knn_search_fp8_rot_timed(const float* src, const float* y_rot, float c0, int N, int T, const void* lib_f8_rot, const void* lib_bf16,
                                              const float* rows_f32, const float* norms, int64_t M, int64_t idx_base, int k, float* out_val,
                                              int32_t* out_idx, void* ws, void* stream, void* ev_start, void* ev_stop) {
    ALIVE_CHECK_ARG(y_rot != nullptr, "alive_knn_search_fp8_rot: null rotated frames");
    return knn_search_fp8_impl(src, N, T, lib_f8_rot, lib_bf16, rows_f32, norms, M, idx_base, k, out_val, out_idx, ws, stream,
                               (hipEvent_t)ev_start, (hipEvent_t)ev_stop, 0, y_rot, c0);
}

// The same search with the candidate stage on the fp6 MFMA (knn_score6_kernel); lib_f6 from alive_library_pack_fp6.
extern "C" int alive_knn_search_fp6(const float* src, int N, int T, const void* lib_f6, const void* lib_bf16, const float* rows_f32,
                                    const float* norms, int64_t M, int64_t idx_base, int k, float* out_val, int32_t* out_idx,
                                    void* ws, void* stream) {
    return knn_search_fp8_impl(src, N, T, lib_f6, lib_bf16, rows_f32, norms, M, idx_base, k, out_val, out_idx, ws, stream, nullptr, nullptr, 2);
}
extern "C" int alive_knn_search_fp6_timed(const float* src, int N, int T, const void* lib_f6, const void* lib_bf16, const float* rows_f32,
                                          const float* norms, int64_t M, int64_t idx_base, int k, float* out_val, int32_t* out_idx,
                                          void* ws, void* stream, void* ev_start, void* ev_stop) {
    return knn_search_fp8_impl(src, N, T, lib_f6, lib_bf16, rows_f32, norms, M, idx_base, k, out_val, out_idx, ws, stream,
                               (hipEvent_t)ev_start, (hipEvent_t)ev_stop, 2);
}

// device pointer (inside ws) to the counters of the last search on this workspace, int[16] (ST_WORDS; slots used today: 0-4, 7-12):
//   [0] frames the fp8 certificate sent to the bf16 stage (all frames when the probe chose bf16 first)
//   [1] frames the bf16 certificate sent on (to the collect tier; [8] of them end in the exact scan)
//   [2] frames of the probe sample, [3] of which failed the fp8 certificate, [4] 1 = the probe chose bf16 first
//   [7] the path taken: 1 streaming scan, 2 exact scan of every frame (k > 8), 3 bf16 first, 4 fp8 first
//   [8] frames the collect tier could not hold (dense clusters) -> exact scan;  [1] then counts the frames sent to the collect tier
//       (strict search with a lo-plane library: the collect tier IS the split-bf16 pass)
// (the counters are the first thing in the workspace: their address depends on neither the batch nor k)
extern "C" const int* alive_knn_search_stats(int N, int T, int64_t M, void* ws) {
    return ws_layout(ws, (int64_t)N * T, M, 4).stats;
}

extern "C" int alive_knn_merge_gather(const float* cand_val, const int32_t* cand_idx, int n_shards, int k, double alpha,
                                      const float* rows_f32_full, const float* src, int N, int T, float* out,
                                      int32_t* final_idx, void* stream) {
    ALIVE_CHECK_ARG(cand_val && cand_idx && rows_f32_full && src && out, "alive_knn_merge_gather: null pointer");
    ALIVE_CHECK_ARG(k >= 1 && k <= ALIVE_MAX_K && n_shards >= 1 && n_shards * k <= 512,
                    "alive_knn_merge_gather: n_shards*k = %d exceeds 512", n_shards * k);
    ALIVE_CHECK_ARG(N > 0 && T > 0, "alive_knn_merge_gather: empty source");
    // with only a few blocks of frames the 12 feature slabs go to separate blocks (streaming: 1 block -> 12)
    const int zs = (int64_t)cdiv(T, 32) * N < 64 ? D / 64 : 1;
    const dim3 g(cdiv(T, 32), N, zs);
    if (k <= 8 && n_shards * k <= 128)
        knn_merge_gather_kernel<2, 8><<<g, 256, 0, (hipStream_t)stream>>>(cand_val, cand_idx, n_shards, k, (float)alpha,
                                                                       (float)(1.0 - alpha), rows_f32_full, src, T, (int64_t)N * T, out, final_idx);
    else
        knn_merge_gather_kernel<8, ALIVE_MAX_K><<<g, 256, 0, (hipStream_t)stream>>>(cand_val, cand_idx, n_shards, k, (float)alpha,
                                                                                 (float)(1.0 - alpha), rows_f32_full, src, T, (int64_t)N * T, out, final_idx);
    ALIVE_CHECK_LAUNCH("alive_knn_merge_gather");
    return ALIVE_OK;
}

extern "C" int alive_dedup_pass(const float* val, const int32_t* idx, int64_t M, int k, double threshold, int32_t* state,
                                int32_t* undecided, void* stream) {
    ALIVE_CHECK_ARG(val && idx && state && undecided && M >= 1 && k >= 1, "alive_dedup_pass: bad arguments");
    dedup_pass_kernel<<<(unsigned)((M + 255) / 256), 256, 0, (hipStream_t)stream>>>(val, idx, M, k, (float)threshold, state, undecided);
    ALIVE_CHECK_LAUNCH("alive_dedup_pass");
    return ALIVE_OK;
}
