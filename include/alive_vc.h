/*
 * alive_vc.h -- C ABI of libalive_vc.so (MI355X / gfx950 voice-conversion hot path).
 *
 * The reference (uthree/ALiVE-VC) is pure PyTorch and has no FFI of its own;
 * its boundary is the Python call surface of its module package.  This header is the
 * native surface that the alive-vc_amd/module package binds with ctypes to implement
 * that Python surface.  Each entry point names the reference function it
 * replaces.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()); fp32 unless noted
 *   - activations are [N][C][T], T contiguous (the reference's layout)
 *   - `stream` is a hipStream_t passed as void*
 *   - no allocation, no synchronisation, no host<->device copy inside: any
 *     call sequence can be captured into a hipGraph.  Scratch comes from the
 *     caller via the *_workspace_bytes queries.
 *   - return 0 on success, negative on error; alive_last_error() gives a
 *     thread-local message.
 */
#ifndef ALIVE_VC_H
#define ALIVE_VC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALIVE_OK 0
#define ALIVE_ERR_ARG (-1)
#define ALIVE_ERR_LAUNCH (-2)

#define ALIVE_DIM 768          /* content feature width (voice_library.py:7) */
#define ALIVE_KPRIME 16        /* candidates kept per frame and library split by the bf16 stage (the fp8 stage keeps 32) */
#define ALIVE_MAX_K 64         /* largest k accepted by the kNN entry points (the reference takes any k <= M, inference.py:34 /
                                * common.py:105).  k <= 8 runs through the MFMA candidate stages (a half-list of the bf16
                                * stage is 8 deep, and all k true neighbours may fall into one of them); 8 < k <= 64 runs the
                                * exact fp32 scan for every frame                                              */

const char* alive_last_error(void);
int alive_version(void);

/* ---------------------------------------------------------------- kNN ----
 * Replaces match_features (module/common.py:96-109) and VoiceLibrary.match
 * (module/voice_library.py:15-33).
 *
 * alive_library_pack: tokens[D][M] (the on-disk layout of voice_library.pt,
 *   generate_voice_library.py:42, batch dim dropped) ->
 *     lib_bf16[M_pad][D]  L2-normalised rows, bf16 (MFMA scoring operand;
 *                         M_pad = alive_library_padded_rows(M), pad rows zero)
 *     rows_f32[M][D]      raw rows, fp32 (exact rescoring + gather)
 *     norms[M]            fp32 L2 norm of each row
 */
int64_t alive_library_padded_rows(int64_t M);
int alive_library_pack(const float* tokens_DxM, int64_t M, int D,
                       void* lib_bf16, float* rows_f32, float* norms, void* stream);

/* alive_knn_search: exact top-k of one library shard, bf16 MFMA candidate stage first.
 *   src[N][D][T] fp32 source frames; frames are flattened to Tt = N*T.
 *   bf16 MFMA cosine scoring with lane-private top-k' lists in LDS, then exact fp32 rescoring (normalise-then-dot, as
 *   the reference) of every candidate.  Every frame is then CERTIFIED: a row outside its rescored set has a candidate
 *   score <= c (the floor of the partial list it failed to enter, or the best candidate the selection of 64 dropped),
 *   hence an exact cosine <= c - min(mu, 0) + max(7 sigma, 2 max|e|), with mu / sigma / max|e| the mean, the RMS about zero
 *   (floored by the stage's typical error) and the largest magnitude of the candidate-score error measured on that frame's
 *   own rescored candidates; the frame passes if its k-th exact cosine clears that.  This certificate is STATISTICAL: it is
 *   wrong for a frame only if the stage error of one of its k true neighbours lies beyond those 7 sigma (audited:
 *   tools/knn_audit.py, profiles/r03_knn_audit.json; alive_knn_search_strict is the form without any such assumption).
 *   Frames that do not pass go through the COLLECT tier: the same bf16 scoring pass with a fixed per-frame threshold (the
 *   frame's k-th exact cosine so far minus the slack it was tested with) that keeps EVERY row at or above it for exact
 *   rescoring -- final unless more such rows exist than its lists hold (clusters of near-copies), and only those frames are searched
 *   again by the exact tier inside the same call: a brute-force fp32 scan of the whole shard with the rescoring
 *   arithmetic (launched up front, sized on the device, no sync).  k > 8: the exact scan for every frame.
 *   out_val[Tt][k] fp32 cosine, descending; out_idx[Tt][k] = idx_base + row.
 *   ws: alive_knn_workspace_bytes(Tt, M) bytes -- the bound that is sufficient for EVERY search entry point of this header, the strict
 *   search with a lo-plane library included (that entry point takes no size argument, so the general query must cover it).
 */
size_t alive_knn_workspace_bytes(int64_t Tt, int64_t M);
/* the same bound under its explicit name: alive_knn_search_strict with lib_lo != NULL keeps both bf16 planes of the frames
 * (2 x 1.5 KB per frame) for its split-bf16 collect tier */
size_t alive_knn_workspace_bytes_strict(int64_t Tt, int64_t M);
/* the smaller workspace of every search WITHOUT a lo-plane library (alive_knn_search, _fp8, _fp6, alive_knn_search_strict with
 * lib_lo == NULL): 3 KB per frame less.  Never hand a buffer of this size to alive_knn_search_strict with lib_lo != NULL. */
size_t alive_knn_workspace_bytes_fast(int64_t Tt, int64_t M);
int alive_knn_search(const float* src, int N, int T,
                     const void* lib_bf16, const float* rows_f32, const float* norms,
                     int64_t M, int64_t idx_base, int k,
                     float* out_val, int32_t* out_idx, void* ws, void* stream);

/* Strict form of alive_knn_search: the same bf16 candidate stage, but the certificate is DETERMINISTIC -- a row outside a
 * frame's rescored set has an exact cosine <= c + || q^ - bf16(q^) || + 1.004 max_R || r^ - bf16(r^) || + 1.0e-4
 * (Cauchy-Schwarz on the two rounding-error vectors; the last term bounds the fp32 accumulation of the stage and the
 * rounding of the rescoring arithmetic).  Frames that do not clear it go through the exact fp32 scan, so the returned
 * lists are the exact top-k of the rescoring arithmetic for EVERY input, adversarial ones included.
 *   bound: device float[1] = max_R || r^ - bf16(r^) ||, filled by alive_library_rounding_bound from a packed library.
 *   lib_lo: NULL, or the library's lo plane (alive_library_pack_lo: bf16(r^ - lib_bf16), 2 * 768 * alive_library_padded_rows(M)
 *           bytes).  With it the frames that fail the certificate (more than 256 of them) go through an MFMA COLLECT pass on BOTH
 *           planes of both operands (three bf16 products per product) whose deterministic bound is 3.0e-4 (two-plane rounding
 *           3 x 2^-18 + the fp32 accumulation of 3 x 768 products, priced for a truncating adder) instead of the single-plane collect tier, whose band is the
 *           certificate's own 1.8e-3: every row at or above v_k - 3.0e-4 is rescored exactly, and only frames with more such rows
 *           than the lists hold reach the exact scan.  ws must then hold alive_knn_workspace_bytes_strict(Tt, M) bytes.
 *   ev_start / ev_stop: as in the *_timed forms below (NULL: none).  Counters: alive_knn_search_stats [1], [7], [8]. */
int alive_library_rounding_bound(const void* lib_bf16, const float* rows_f32, const float* norms, int64_t M,
                                 float* bound, void* stream);
int alive_library_pack_lo(const void* lib_bf16, const float* rows_f32, const float* norms, int64_t M, void* lib_lo, void* stream);
int alive_knn_search_strict(const float* src, int N, int T,
                            const void* lib_bf16, const void* lib_lo, const float* rows_f32, const float* norms, const float* bound,
                            int64_t M, int64_t idx_base, int k,
                            float* out_val, int32_t* out_idx, void* ws, void* stream, void* ev_start, void* ev_stop);

/* The same search with the first candidate stage on the block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, OCP e4m3
 * operands = normalised rows x 2^8, scales 2^0): about twice the scoring rate at ~20x the score error of bf16, so the
 * lists hold twice as many candidates (32 per frame and library split) in front of the same exact fp32 rescoring.
 *   lib_f8[M_pad][D]: alive_library_fp8_bytes(M) bytes, made from lib_bf16 by alive_library_pack_fp8.
 *   Same workspace, same outputs and the same contract as alive_knn_search.
 * Tiers, all launched up front and decided on the device (no sync, graph-capturable):
 *   probe  batches of >= 16384 frames: the fp8 stage and its certificate on a sample of 1024 frames; when more than 55 %
 *          of the sample fail (a library whose best cosines lie closer together than the fp8 error) the fp8 pass over
 *          the batch is skipped and every frame starts at the bf16 stage;
 *   fp8    candidates, exact rescoring, certificate (fp8 error statistics).  Batches of >= 512 x 256 frames: the blocks of
 *          library split s start their candidate lists at the seeds split s - 1 left for their frames (k-th best fp8 score seen so
 *          far minus 0.02: rows below it are never admitted, and the certificate counts the seed as the bound on them);
 *   bf16   the frames that failed (up to 64 of them: straight to the exact scan), compacted, through the bf16 stage:
 *          candidates, rescoring, certificate (bf16 statistics);
 *   collect the frames that failed again: bf16 scoring pass with a fixed threshold, every row above it rescored exactly;
 *   exact  the frames whose collected rows overflowed: brute-force fp32 scan.
 * alive_knn_search_stats: device pointer (inside ws) to int[16] counters of the last search on that workspace:
 *   [0] frames that failed the fp8 certificate (<= 64: exact scan; more: the bf16 stage)
 *   [1] frames that failed the bf16 certificate (<= 256: exact scan; more: the collect tier)  [8] frames the collect tier
 *   sent on to the exact scan  [2] probe sample size  [3] probe failures  [9] / [10] fp8 / bf16 blocks that started from seeds
 *   [11] / [12] the two limits just named (64, 256), written with the counters
 *   [4] 1 = the probe chose bf16 first  [7] the path taken: 1 streaming scan, 2 exact scan of every frame (k > 8),
 *   3 bf16 first, 4 fp8 first, 5 fp6 first.  (alive_knn_search fills [1] and [7] only.)  The counters sit at the start of ws. */
size_t alive_library_fp8_bytes(int64_t M);
int alive_library_pack_fp8(const void* lib_bf16, int64_t M, void* lib_f8, void* stream);
int alive_knn_search_fp8(const float* src, int N, int T,
                         const void* lib_f8, const void* lib_bf16, const float* rows_f32, const float* norms,
                         int64_t M, int64_t idx_base, int k,
                         float* out_val, int32_t* out_idx, void* ws, void* stream);
const int* alive_knn_search_stats(int N, int T, int64_t M, void* ws);

/* Measurement forms (bench.py): the same searches; ev_start / ev_stop are hipEvent_t handles (or NULL) recorded on `stream`
 * immediately before / after the first-stage scoring kernel of the whole batch -- the dominant kernel of the path. */
int alive_knn_search_timed(const float* src, int N, int T,
                           const void* lib_bf16, const float* rows_f32, const float* norms,
                           int64_t M, int64_t idx_base, int k,
                           float* out_val, int32_t* out_idx, void* ws, void* stream, void* ev_start, void* ev_stop);
int alive_knn_search_fp8_timed(const float* src, int N, int T,
                               const void* lib_f8, const void* lib_bf16, const float* rows_f32, const float* norms,
                               int64_t M, int64_t idx_base, int k,
                               float* out_val, int32_t* out_idx, void* ws, void* stream, void* ev_start, void* ev_stop);

/* Round 6: the fp8 search on ROTATED operands, for DENSE banks (a single speaker: every row shares a large common component, thousands
 * of rows lie inside an 8-bit stage's error of a frame's k-th neighbour and the plain fp6 / fp8 stages certify next to nothing).  The
 * unit vectors are expressed in a basis of the bank's own subspace -- 64 leading principal directions, carried as two e4m3 digits each,
 * and 512 further coordinates mixed by a fixed rotation; /root/reference/module/content_encoder.py ends in Conv1d(512 -> 768), so a bank
 * has rank <= 513 and nothing is lost -- and laid out so that the unchanged scoring kernel's dot product of the two 768-code vectors is
 * the cosine at a stage error of 2.4e-4 instead of 1.6e-3 (profiles/r06_knn_pca_probe.json; csrc/knn.hip rot_codes_kernel).  The
 * caller owns the basis (module/common.py::PackedLibrary builds it when a bank is packed and rotates the frames with alive_conv1d):
 *   alive_knn_rot_coordinates() = 576 coordinates per vector are passed: the first alive_knn_rot_leading() = 64 the leading directions,
 *       the next alive_knn_rot_mixed() = 507 the mixed ones, the last 5 unread;
 *   c0: the bank's centring constant of the leading coordinate (an e4m3-exact value near the mean of the rows' first coordinate: that
 *       coordinate is ~0.65 for every vector of a dense bank and is carried as c0 + two digits of the difference);
 *   alive_library_pack_fp8_rot: y_rows[count][576] = coordinates of the UNIT rows m0 .. m0 + count - 1 -> their codes in lib_f8
 *       (alive_library_fp8_bytes(M) bytes; chunks in ascending order, m0 a multiple of 32, the last chunk ends at M);
 *   alive_knn_search_fp8_rot_timed: alive_knn_search_fp8_timed with y_rot[N][576][T] = W^T src (not normalised) beside src.
 * Results are those of every other search: exact fp32 rescoring on the original rows, the same certificates (stage prior 2.5e-4),
 * the bf16 tiers on lib_bf16 behind them. */
int alive_knn_rot_coordinates(void);
int alive_knn_rot_leading(void);
int alive_knn_rot_mixed(void);
int alive_library_pack_fp8_rot(const float* y_rows, int64_t m0, int64_t count, int64_t M, float c0, void* lib_f8, void* stream);
int alive_knn_search_fp8_rot_timed(const float* src, const float* y_rot, float c0, int N, int T,
                                   const void* lib_f8_rot, const void* lib_bf16, const float* rows_f32, const float* norms,
                                   int64_t M, int64_t idx_base, int k,
                                   float* out_val, int32_t* out_idx, void* ws, void* stream, void* ev_start, void* ev_stop);

/* Round 5: the same tiered search with the first candidate stage on the fp6 form of that MFMA (OCP e2m3 operands on both sides =
 * normalised rows x 2^5, scales 2^0; /root/reference/module/common.py:100-105 is still what comes out).  The matrix pipe runs e2m3
 * at twice the e4m3 rate, and for unit vectors a 1/8 grid is as accurate as three mantissa bits (score error 1.8e-3 in cosine
 * against 1.35e-3): the per-frame certificate is the fp8 stage's with a prior of 2.0e-3, the lists are as deep (32 candidates per
 * frame and library split), frames that fail go to the bf16 stage exactly as above.  A wave keeps 96 frames stationary (216
 * registers) and a block 384, so one 32-byte LDS fragment feeds three MFMAs.
 *   lib_f6[M_pad][D]: alive_library_fp8_bytes(M) bytes from alive_library_pack_fp6 -- every group of 32 features keeps its 32-byte
 *   slot: 32 six-bit codes as a little-endian bit stream in the first 24 bytes, 8 bytes of zeros.
 *   Workspace, outputs, counters (alive_knn_search_stats; [7] = 5: fp6 first) as for alive_knn_search_fp8. */
int alive_library_pack_fp6(const void* lib_bf16, int64_t M, void* lib_f6, void* stream);
int alive_knn_search_fp6(const float* src, int N, int T,
                         const void* lib_f6, const void* lib_bf16, const float* rows_f32, const float* norms,
                         int64_t M, int64_t idx_base, int k,
                         float* out_val, int32_t* out_idx, void* ws, void* stream);
int alive_knn_search_fp6_timed(const float* src, int N, int T,
                               const void* lib_f6, const void* lib_bf16, const float* rows_f32, const float* norms,
                               int64_t M, int64_t idx_base, int k,
                               float* out_val, int32_t* out_idx, void* ws, void* stream, void* ev_start, void* ev_stop);

/* alive_knn_merge_gather: merge n_shards exact top-k lists ([S][Tt][k], e.g.
 * after an RCCL all-gather), pick the global top-k, gather those rows from the
 * full fp32 row table, mean over k, alpha-blend with the source.  n_shards * k <= 512.
 *   out[N][D][T];  final_idx[Tt][k] (may be NULL).
 */
int alive_knn_merge_gather(const float* cand_val, const int32_t* cand_idx, int n_shards, int k,
                           double alpha, const float* rows_f32_full, const float* src,
                           int N, int T, float* out, int32_t* final_idx, void* stream);

/* alive_dedup_pass: one pass of the greedy de-duplication of a library (generate_voice_library.py of this build, --dedup):
 * frame i is dropped iff a KEPT earlier frame among its k nearest (val / idx[M][k] = alive_knn_search of the library
 * against itself) has cosine > threshold.  state[M]: 0 undecided, 1 kept, 2 dropped (zero it before the first pass);
 * *undecided (zero it before each pass) counts the frames still waiting for an earlier neighbour; repeat until it is 0. */
int alive_dedup_pass(const float* val, const int32_t* idx, int64_t M, int k, double threshold, int32_t* state,
                     int32_t* undecided, void* stream);

/* ----------------------------------------------------------- operators ----
 * Building blocks, exported so that every kernel has its own parity test.
 *
 * alive_conv1d: Conv1d / ConvTranspose1d(k == stride) as an f32-MFMA implicit
 * GEMM with fused epilogue.  Replaces every nn.Conv1d / nn.ConvTranspose1d on
 * the path (common.py:48-51,88-92; decoder.py:16-17,41,61,108-110,141,164-182).
 */
typedef struct AliveConv {
    const float* W;        /* packed [Co_pad][K_pad], K = Ci*KW (k-major within ci), zero padded to x16 */
    const float* bias;     /* [rows] or NULL (rows = Co, or Co*up for transposed) */
    const float* X;        /* [N][Ci][Tin] */
    int N, Ci, Tin;
    int Co;                /* GEMM rows (Co, or Co*up for transposed convs) */
    int K_pad;             /* padded K (multiple of 16) */
    int KW, stride, dil;   /* input index = t*stride + j*dil - pad_left */
    int pad_left;
    int pad_mode;          /* 0 zero; 1 reflect left, zero right; 2 reflect both (STFT centre pad) */
    int Tout;              /* GEMM columns per batch item */
    int up;                /* 1, or r for ConvTranspose1d(k=r, stride=r): row -> (co=row/r, j=row%r), t -> t*r+j */
    int act;               /* 0 none, 1 gelu(erf), 2 exp, 3 sin */
    const float* post_add; /* [rows] added after act (FiLM "+1"), or NULL */
    const float* ch_scale; /* [rows] multiplied after act (ConvNeXt layer scale), or NULL */
    const float* residual; /* [N][Co][Tout] added after ch_scale, or NULL */
    const float* skip;     /* [N][Co][Tout] added after residual, or NULL */
    float* Y;              /* raw output [N][Co_out][Tout*up], or NULL */
    /* optional second output: Z = gelu(v) * interp(film[fs]) + interp(film[fh])  (decoder.py:112-117,130-132) */
    float* Z;
    const float* film;     /* [N][film_rows][Lf] */
    int film_rows, Lf, film_scale_row, film_shift_row;
    /* arithmetic: 0 = exact fp32 (f32-input MFMA), W as above;
     *             1 = 2-term split bf16 ("bf16x3": hi*hi + hi*lo + lo*hi on the bf16 MFMA, ~2^-16 per product),
     *                 W = bf16, 2 planes (hi, lo) of Co_pad rows x K = KW*Ci_pad, tap-major k = j*Ci_pad + ci, stored K-BLOCKED
     *                 like the activation planes below: [2][K/32][Co_pad][32] (module/_pack.py::pack_conv_split);
     *                 Ci_pad a multiple of 32; stride must be 1;
     *             2 = 3-term split bf16 ("bf16x6", six MFMAs per product, fp32-grade): W as for 1 with 3 planes;
     *             3 = plain fp16 (round 5): ONE MFMA per product (v_mfma_f32_32x32x16_f16), operands rounded to nearest even and saturated
     *                 at +-65504, fp32 accumulate -- W = ONE fp16 plane [K/32][Co_pad][32] (module/_pack.py::pack_conv_split_h stores it
     *                 as the third slab behind the two bf16 planes), Xp / Zp carry ONE fp16 plane.  2^-12 per operand: for layers whose
     *                 contribution to the output has been measured (alive_decoder_precision).  With at most 96 columns (the streaming
     *                 kernels) it is not available: those callers use precision 1. */
    int precision, Ci_pad;
    /* FiLM of a frame RANGE of a longer window (alive_decoder_forward_range): this conv's columns start at sample
     * film_t0 of the window (at its own rate) and `film` holds the frames [film_f0, film_f0 + film_ld) only, row pitch
     * film_ld; Lf stays the window's frame count, so the interpolation coordinates are those of the whole window.
     * film_ld == 0: film covers the whole window (film_t0 = film_f0 = 0, pitch Lf).  Split kernel only. */
    int film_t0, film_f0, film_ld;
    /* Plane-packed operands of the split kernel (precision 1, Co > 64; round 3).  Format = "plane-packed activations" below:
     * P[plane][C_pad/32][cols_pad][32] bf16 (row = n * T + t), 2 planes, plane stride = cols_pad * C_pad, cols_pad = N * T
     * rounded up to 128.
     *   Xp  input instead of X (X is ignored): Ci a multiple of 32, pad_mode 1 (or no padding); staged by LDS-DMA.
     *   Zp  the second output (gelu + FiLM) in that format instead of / beside Z: Co a multiple of 64, Tout a multiple of 4.
     * The decoder's 256-channel FilterBlock chains its convs through these (networks.hip). */
    const void* Xp;
    void* Zp;
    /* Round 5: Y ALSO as k-blocked bf16 planes (2 planes, [2][pad32(Co) / 32][cols_pad][32], cols = N * Tout), written by the exact
     * fp32 kernel (precision 0) beside Y for a plain conv (no activation / post_add / ch_scale / residual / skip / Z, up 1,
     * 16 < Co <= 64, Co % 4 == 0): the plane image of the Filter's 64-channel skip tensor comes from the conv that produces it
     * (decoder.py:186-188) instead of an alive_to_planes pass over it.  Same bits as alive_to_planes(Y). */
    void* Yp;
    int yp_planes;         /* 0 / 2: two bf16 planes (the bits of alive_to_planes(Y, 2)); 1: ONE fp16 plane (alive_to_planes(Y, 1)) */
} AliveConv;
int alive_conv1d(const AliveConv* desc, void* stream);

/* ---- plane-packed activations: the frame-rate GEMMs of the ConvNeXt stacks --------------------------
 * The 1x1 convs of ContentEncoder / F0Estimator / FeatureExtractor (content_encoder.py:22-25,
 * f0_estimator.py:23-27, decoder.py:43-48, common.py:57-61,77-81) run as plain GEMMs whose activation operand is
 * already split into bf16 planes and stored K-BLOCKED, so that BOTH operands go global -> LDS by LDS-DMA in contiguous
 * runs of whole cache lines and the inner loop is MFMAs only (csrc/planes_layout.h):
 *   P[plane][C_pad/32][cols_pad][32] bf16 -- element (plane, col, c) at ((plane * C_pad/32 + c/32) * cols_pad + col) * 32 + c % 32,
 *   col = n*T + t, plane 0 = bf16(v), plane 1 = bf16(v - p0), plane 2 = bf16(v - p0 - p1);
 *   C_pad a multiple of 32 (zero filled), plane stride = cols_pad * C_pad with cols_pad a multiple of 128.
 *   (Rounds 2 - 3 stored the planes row-major, [plane][col][C_pad]: a 16-row DMA piece then touched 16 half lines whose other
 *   halves the next K-step fetched again; the k-blocked form took 5 - 12 % off every kernel that reads them.)
 * alive_to_planes converts an fp32 [N][C][T] tensor; alive_gemm_planes writes fp32 [N][Co][T] (same epilogue
 * options as alive_conv1d) and / or plane-packed output (act applied first) for the next GEMM. */
size_t alive_planes_bytes(int64_t cols, int C, int planes);        /* bytes of a plane-packed buffer */
int alive_to_planes(const float* X, int N, int C, int T, int planes, void* P, void* stream);
typedef struct AliveGemm {
    const void* W;         /* bf16 [planes][Ci_pad/32][Co_pad][32], k-blocked like P (module/_pack.py::pack_conv_split of a k=1 conv) */
    const float* bias;     /* [Co] or NULL */
    const void* P;         /* input planes [planes][Ci_pad/32][cols_pad][32] */
    int N, T;              /* cols = N*T; fp32 outputs are [N][Co][T] */
    int Ci, Co;
    int planes;            /* 2: bf16x3, 3: bf16x6 (both operands); 1 (round 5): plain fp16, one MFMA per product -- W, P and Pout are
                            * single fp16 planes in the same k-blocked layout; act 0 - 2 only */
    int act;               /* 0 none, 1 gelu, 2 exp, 3 argmax over Co (3 planes only): no Y / Pout, see arg_val; 4: see the end of the struct */
    const float* post_add; /* [Co] or NULL */
    const float* ch_scale; /* [Co] or NULL */
    const float* residual; /* [N][Co][T] or NULL */
    float* Y;              /* fp32 output [N][Co][T] or NULL */
    void* Pout;            /* plane-packed output [planes][Co_pad32/32][cols_pad][32] or NULL (Co_pad32 = Co rounded up to 32) */
    /* custom placement of the input rows (all 0: the packed form above).  Row (n, t) of plane pl starts at element
     * pl*b_plane + n*b_win + t*b_row: overlapping rows (b_row < Ci) make the GEMM a strided convolution over a signal
     * stored once -- the STFT runs this way (b_row = hop 320, Ci = 1280).  Multiples of 8 elements. */
    int64_t b_plane, b_win;
    int b_row;
    /* b_cblk = 0: such a row is k-contiguous in memory (the STFT's signal).  b_cblk > 0: the rows live in a k-blocked plane buffer
     * (alive_to_planes / Pout) of b_cblk k-blocks, and the K vector of an output column is Ci / (32 b_cblk) consecutive rows of it:
     * k = (tap * b_cblk + block) * 32 + i is element i of row + tap in block `block`, b_blk elements per block (the padded rows of
     * that buffer * 32) -- the Filter's strided down convs (k == stride) run this way on the output planes of the layer before. */
    int b_cblk;
    int64_t b_blk;
    /* act == 3: the [Co][cols] product is never stored.  Each 64-row block of the GEMM leaves, per column, its largest
     * value (bias included) and the row it sits in -- arg_val / arg_idx [ceil(Co/64)][N*T], first row on ties, NaN wins
     * like ATen -- and alive_argmax_merge reduces the blocks: F0Estimator.estimate (f0_estimator.py:30-34) without the
     * 4096-class logits tensor. */
    float* arg_val;
    int32_t* arg_idx;
    /* Round 5 (the front end without an fp32 spectrogram).  Fields appended: a zeroed struct of the old size means "off".
     * y_split > 0: ONE GEMM for two consumers of the same input -- rows [0, y_split) go to Y as [N][y_split][T], rows [y_split, Co) to
     *   Y2 as [N][Co - y_split][T] (y_split a multiple of 128; no residual).  The ContentEncoder / F0Estimator input layers
     *   (content_encoder.py:22, f0_estimator.py:23) run as one 641 -> 512 + 256 GEMM this way.
     * act == 4: the rows are (re, im) PAIRS of a DFT (row 2f = cos, 2f + 1 = -sin: alive_dft_basis's interleaved image); the epilogue
     *   takes |re + i im| (hypotf, spectrogram.py:8) of each pair and leaves the Co / 2 magnitudes as plane-packed channels in Pout
     *   ([planes][pad32(Co / 2) / 32][cols_pad][32]) -- the magnitude spectrogram never exists in fp32.  No Y, bias or other term. */
    float* Y2;
    int y_split;
    /* Round 5: fp16 split planes (planes == 2 only; fields appended, all 0 = the bf16 planes above).  f16s != 0: both operands are TWO
     * fp16 planes of a power-of-two multiple of the values -- hi = fp16(s v), lo = fp16(s v - hi), round to nearest even, saturated at
     * +-65504 -- in the same k-blocked layout, multiplied as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16: 22 significand bits per
     * operand (the two bf16 planes carry 16) for elements above 2^-11 / s, an absolute error below 2^-25 / s for smaller ones -- fp32-grade
     * at three MFMAs per product instead of the six of the three-plane bf16 form.  The accumulator is multiplied by *wscale * in_unscale
     * before the bias (wscale: device pointer to 1 / s_W of this weight tensor, module/_pack.py::pack_conv_split_f16s; in_unscale =
     * 1 / s_P, the scale the producer of P used); Pout, if set, is written in the same format with the scale pout_scale.
     * Used by the ConvNeXt pointwise convs of both encoders (alive_encoder_precision). */
    int f16s;
    const float* wscale;
    float in_unscale, pout_scale;
} AliveGemm;
int alive_gemm_planes(const AliveGemm* desc, void* stream);
/* out[col] = (float) row of the largest value over the nblk per-block candidates of column col (smallest row on ties) */
int alive_argmax_merge(const float* arg_val, const int32_t* arg_idx, int nblk, int64_t cols, float* out, void* stream);

/* FilterBlock.forward (decoder.py:137-150) for C = 8 or 16 fused into one kernel (filter_small.hip): input_conv 1x1 + three
 * FilterResBlocks (six GELU -> FiLM -> reflect-left causal k5 convs, dilations 1,1,2,2,4,4), optional U-Net skip
 * added to the result.  U[N][C][L] -> out[N][C][L] (not in place).
 *   wpack: alive_filter_block_small_weights(C) floats, 16-byte aligned = fp32 biases [7][32] (input conv first, rows >= C zero), then
 *          bf16 pairs in fp32 words: input conv [2 planes][32 rows][16] (k = ci) and per conv q = 0..5 [2 planes][32 rows][KP],
 *          k = tap * C + ci, KP = 80 (C = 16) / 48 (C = 8), zero padded; planes hi = bf16(w), lo = bf16(w - hi)
 *          (module/_pack.py::pack_filter_small).  Since round 4 the block computes on the split-bf16 product of the 32x32x16 MFMA
 *          (~2^-16 per product, like the 64- and 256-channel scales), not on the exact f32 MFMA
 *   film[N][film_rows][Lf]; conv q reads its scale rows at film_off + q*2C and shift rows at film_off + q*2C + C. */
int alive_filter_block_small_weights(int C);
int alive_filter_block_small(const float* U, int N, int C, int L, const float* wpack, const float* film,
                             int film_rows, int Lf, int film_off, const float* skip, float* out, void* stream);
/* the same on a frame range of a longer window (see AliveConv.film_t0): L samples starting at sample t0 of the window's
 * tensor at this rate, film[N][film_rows][film_ld] holding the frames from f0 on, Lf = frames of the whole window */
int alive_filter_block_small_range(const float* U, int N, int C, int L, const float* wpack, const float* film,
                                   int film_rows, int Lf, int film_off, int t0, int f0, int film_ld, const float* skip,
                                   float* out, void* stream);

/* FilterBlock.forward (decoder.py:137-150) for C = 64 fused into one kernel on the split-bf16 MFMA (filter_mid.hip):
 * activations stay in LDS as two bf16 planes through the input conv and the six modulated k5 convs.
 *   W16: bf16, alive_filter_block64_weights() elements = input conv [2 planes][64][64], then per conv q = 0..5
 *        [2 planes][64][k = j*64 + ci]  (module/_pack.py::pack_filter_mid)
 *   biases: fp32 [7][64], input conv first;  film / film_off / skip / out as alive_filter_block_small;
 *   L a multiple of 4, out and skip 16-byte aligned, not in place. */
int64_t alive_filter_block64_weights(void);
int alive_filter_block64(const float* U, int N, int L, const void* W16, const float* biases, const float* film,
                         int film_rows, int Lf, int film_off, const float* skip, float* out, void* stream);
int alive_filter_block64_range(const float* U, int N, int L, const void* W16, const float* biases, const float* film,
                               int film_rows, int Lf, int film_off, int t0, int f0, int film_ld, const float* skip,
                               float* out, void* stream);
/* the same with the six k5 convs on ONE fp16 plane per operand (one MFMA per product, v_mfma_f32_32x32x16_f16; the 1x1 input conv keeps
 * split bf16): W16 = the pack of alive_filter_block64_weights() elements, whose last 6 x 64 x 320 hold the k5 weights as fp16
 * (module/_pack.py::pack_filter_mid).  Part of alive_decoder_precision mode 1. */
int alive_filter_block64_range_fp16(const float* U, int N, int L, const void* W16, const float* biases, const float* film,
                               int film_rows, int Lf, int film_off, int t0, int f0, int film_ld, const float* skip,
                               float* out, void* stream);

/* FilterBlock.forward (decoder.py:137-150) for C = 256 fused into one kernel on plain fp16 operands (filter_big.hip; decoder precision
 * mode 1, batch path): a block sweeps a segment of a window in tiles of 128 columns, the six modulated k5 convs' inputs stay in LDS as one
 * fp16 plane, the residual stream in registers.  The 1x1 input conv is part of the transposed conv that produces U (module/_pack.py).
 *   w16[q], bias[q], q = 0..5: blocks[q / 2].c1 / .c2 -- the fp16 slab [K / 32][256][32] (K = 5 x 256, tap-major) of
 *        module/_pack.py::pack_conv_split_h (third slab of the pack) and the fp32 bias [256];  w16 and bias are HOST arrays of device pointers.
 *   film / film_off / t0 / f0 / film_ld / skip / out as alive_filter_block64_range;  L > 32;  not in place;
 *   ws: alive_filter_block256_workspace_bytes(N, L) bytes of device scratch (the convs' causal contexts between a block's tiles). */
int64_t alive_filter_block256_workspace_bytes(int N, int L);
int alive_filter_block256_fp16(const float* U, int N, int L, const void* const* w16, const float* const* bias, const float* film,
                               int film_rows, int Lf, int film_off, int t0, int f0, int film_ld, const float* skip, float* out,
                               void* ws, int64_t ws_bytes, void* stream);

/* the same kernel for C = 64 (decoder scale 1; round 6): tiles of 512 columns, four column groups of waves, FiLM at <= 1 frame per 25
 * columns.  Like the 256-channel form it takes the block's residual stream -- the decoder composes blocks[1].input_conv into ups[1]
 * (module/_pack.py "flt.up1.Wc") -- and six fp16 slabs [K / 32][64][32], K = 5 x 64.  alive_filter_block64_range_fp16 (input conv inside,
 * one bf16-typed pack) stays for the split-bf16 mode and for short signals. */
int64_t alive_filter_block64s_workspace_bytes(int N, int L);
int alive_filter_block64s_fp16(const float* U, int N, int L, const void* const* w16, const float* const* bias, const float* film,
                               int film_rows, int Lf, int film_off, int t0, int f0, int film_ld, const float* skip, float* out,
                               void* ws, int64_t ws_bytes, void* stream);

/* The waveform-rate edges of Filter.forward (decoder.py:164,182,186-188,194) as streaming kernels:
 *   alive_filter_source_in : downs[0](source_in(src)):  src[N][Lw] -> d0[N][16][Lw/2]
 *                            Win[8][7], bin[8] = source_in (pad 3); Wd[16][8][2], bd[16] = downs[0] (stride 2); fp32,
 *                            reference layouts.  The 8-channel intermediate is not a skip and is never stored.
 *   alive_filter_source_out: source_out(H):  H[N][8][Lw] -> wave[N][Lw];  W[8][7] (= weight[0]), b[1] */
int alive_filter_source_in(const float* src, int N, int Lw, const float* Win, const float* bin, const float* Wd,
                           const float* bd, float* d0, void* stream);
int alive_filter_source_out(const float* H, int N, int Lw, const float* W, const float* b, float* wave, void* stream);

/* depthwise k7 conv + (Adaptive)ChannelNorm (common.py:20-26,35-41,55-56,75-76)
 *   affine_mode 0: gain[C], offset[C];  1: per-sample scale/shift rows in cond[N][cond_rows][T] */
int alive_dwconv_norm(const float* X, int N, int C, int T, const float* dw_w, const float* dw_b,
                      int affine_mode, const float* gain, const float* offset,
                      const float* cond, int cond_rows, int scale_row, int shift_row,
                      float eps, float* Y, void* stream);
/* the same with plane-packed output for alive_gemm_planes (dw_w == dw_b == NULL: the norm alone); one pass over memory:
 * the [C][64 columns] tile stays in LDS between the conv, the statistics, the affine and the split.  C a multiple of 32. */
int alive_dwconv_norm_planes(const float* X, int N, int C, int T, const float* dw_w, const float* dw_b,
                             int affine_mode, const float* gain, const float* offset,
                             const float* cond, int cond_rows, int scale_row, int shift_row,
                             float eps, int planes, void* P, void* stream);
/* the same with fp16 split planes: TWO planes hi = fp16(s y), lo = fp16(s y - hi) with s = scale (a power of two), saturated at
 * +-65504 -- the input format of alive_gemm_planes with f16s (AliveGemm.in_unscale = 1 / scale) */
int alive_dwconv_norm_planes_f16s(const float* X, int N, int C, int T, const float* dw_w, const float* dw_b,
                                  int affine_mode, const float* gain, const float* offset,
                                  const float* cond, int cond_rows, int scale_row, int shift_row,
                                  float eps, float scale, void* P, void* stream);
/* z = gelu(h) * interp(film[scale_row + c]) + interp(film[shift_row + c])  (decoder.py:112-117,130-132: F.gelu, then the FiLM of a
 * ModulatedCausalConv1d with F.interpolate(mode='linear')), the arithmetic of alive_conv1d's second output.  H [N][C][L];
 * film [N][film_rows][film_ld] holds the frames from f0 on of a window of Lf frames, t0 = first sample of H in the window at this
 * rate (whole window: t0 = f0 = 0, film_ld = Lf).  Exactly one of Z (fp32 [N][C][L]) and Zp (2 k-blocked bf16 planes, C % 32 == 0). */
int alive_gelu_film(const float* H, int N, int C, int L, const float* film, int film_rows, int Lf, int scale_row,
                    int shift_row, int t0, int f0, int film_ld, float* Z, void* Zp, void* stream);
/* ChannelNorm alone (f0_estimator.py:25) */
int alive_channel_norm(const float* X, int N, int C, int T, const float* gain, const float* offset,
                       float eps, float* Y, void* stream);
/* argmax over channels -> float (f0_estimator.py:33) */
int alive_argmax_channels(const float* X, int N, int C, int T, float* out, void* stream);

/* HarmonicOscillator.forward after to_amps/exp (decoder.py:79-100).
 *   amps[N][H][Lf] (already exp'd), f0[N][Lf], phi_in[N][H] or NULL, crop0,
 *   wave[N][Lf*seg], phi_out[N][H] = asin(sin(theta)) at column phi_col (or NULL).
 *   ws: alive_oscillator_workspace_bytes(N, H, Lf). */
size_t alive_oscillator_workspace_bytes(int N, int H, int Lf);
int alive_oscillator(const float* amps, const float* f0, const float* phi_in, int N, int H, int Lf,
                     int seg, float sample_rate, int crop0, int phi_col,
                     float* wave, float* phi_out, void* ws, void* stream);
/* the frames [f_begin, f_begin + n_frames) of the window only: amps[N][H][n_frames], wave[N][n_frames*seg]; f0 and the
 * phase accumulation cover the whole window, so the samples are bitwise those of alive_oscillator */
int alive_oscillator_range(const float* amps, const float* f0, const float* phi_in, int N, int H, int Lf,
                           int seg, float sample_rate, int crop0, int phi_col, int f_begin, int n_frames,
                           float* wave, float* phi_out, void* ws, void* stream);

/* magnitude STFT 1280/320, rect window, reflect centre pad, last frame dropped
 * (module/spectrogram.py:5-10): wav[N][L] -> spec[N][641][L/320], as a DFT GEMM on the
 * f32 MFMA.  basis: [1296][1280] fp32 filled once by alive_dft_basis (rows 0..640 cos,
 * 641..1281 -sin, rest zero); ws: alive_spectrogram_workspace_bytes(N, L). */
size_t alive_dft_basis_bytes(void);
int alive_dft_basis(float* basis, void* stream);
size_t alive_spectrogram_workspace_bytes(int N, int L);
int alive_spectrogram(const float* basis, const float* wav, int N, int L, float* spec, void* ws, void* stream);

/* Decoder.forward on the frames [f_begin, f_begin + n_frames) of windows of Lf frames (context trimming: inference.py
 * keeps the centre third of a window).  x[N][768][n_frames] are the matched features of that range, f0[N][Lf] covers the
 * whole window (phase accumulation), wave[N][320*n_frames].  Frames further than the decoder's receptive field from the
 * range edges (12 + 12 to the left, 13 to the right) are bitwise those of alive_decoder_forward on the whole window.
 * ws: alive_decoder_workspace_bytes(N, n_frames) + alive_decoder_workspace_bytes(N, Lf) is enough. */
int alive_decoder_forward_range(const float* const* w, const float* x, const float* f0, int N, int Lf, int f_begin,
                                int n_frames, float* wave, void* ws, void* stream);

/* ------------------------------------------------------- audio edges (f2) ----
 * torchaudio.functional.resample (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99) as a polyphase filter bank,
 * torchaudio.functional.gain folded in as a pre / post factor, and the int16 conversions of the streaming loop
 * (inference.py:88-94,135-142; realtime_inference.py:139-147,173-183).  orig / new are the rates divided by their gcd.
 *   filt: [new][alive_resample_taps(orig, new)] floats filled once by alive_resample_filter;
 *   y[b][o] = post_scale * sum_j filt[o % new][j] * (pre_scale * x[b][(o / new) * orig + j - width]), zero outside x;
 *   Lout <= alive_resample_length(L, orig, new) = ceil(new * L / orig).
 *   alive_float_to_pcm16 truncates toward zero and keeps the low 16 bits (numpy astype, no clipping). */
int alive_resample_taps(int orig, int new_rate);
int64_t alive_resample_length(int64_t L, int orig, int new_rate);
int alive_resample_filter(int orig, int new_rate, float* filt, void* stream);
int alive_resample(const float* x, int B, int L, int orig, int new_rate, const float* filt, float pre_scale,
                   float post_scale, float* y, int Lout, void* stream);
int alive_pcm16_to_float(const int16_t* in, int64_t n, float* out, void* stream);
int alive_float_to_pcm16(const float* in, int64_t n, int16_t* out, void* stream);

/* ------------------------------------------------------------ networks ----
 * Weight tables are arrays of device pointers in the order given by
 * alive_weight_name(model, i), i < alive_weight_count(model); tensors are the
 * packed forms produced by module/_pack.py from a reference state_dict.
 * model: 0 content encoder, 1 f0 estimator, 2 decoder.
 */
int alive_weight_count(int model);
const char* alive_weight_name(int model, int index);

/* ContentEncoder.forward (content_encoder.py:21-25): spec[N][641][T] -> out[N][768][T] */
/* Fused front end (round 5): wav[N][L] at 16 kHz -> content features feat[N][768][L / 320] and f0 classes
 * f0[N][1][L / 320] (the argmax of F0Estimator.estimate, before the class -> Hz map is applied by the caller exactly as after
 * alive_f0_estimate), i.e. spectrogram.py:5-10 + content_encoder.py:21-25 + f0_estimator.py:22-34 in one call and without an
 * fp32 spectrogram: bitwise alive_spectrogram + alive_f0_estimate + alive_content_encoder.  Batch path only (N * (L / 320) >= 96,
 * L % 8 == 0).  w_in: the two input layers' plane-packed weights concatenated along the rows (CE's 512, then PE's 256:
 * [3][672 / 32][768][32] bf16), b_in: their biases [768].  ws: alive_front_end_workspace_bytes(N, L). */
size_t alive_front_end_workspace_bytes(int N, int L);
int alive_front_end(const float* basis, const float* const* ce_weights, const float* const* pe_weights, const void* w_in,
                    const float* b_in, const float* wav, int N, int L, float* feat, float* f0, void* ws, void* stream);
size_t alive_content_encoder_workspace_bytes(int N, int T);
int alive_content_encoder(const float* const* w, const float* spec, int N, int T,
                          float* out, void* ws, void* stream);

/* F0Estimator.estimate (f0_estimator.py:29-34): spec -> f0[N][T] (class index as float) */
size_t alive_f0_estimate_workspace_bytes(int N, int T);
int alive_f0_estimate(const float* const* w, const float* spec, int N, int T,
                      float* f0, void* ws, void* stream);

/* Arithmetic of the decoder's two largest groups of GEMMs on the batch path (more than 96 columns; the streaming kernels are not
 * affected): (a) the six k = 5 convs of the 256-channel FilterBlock (decoder.py:128-134 at the Filter's coarsest scale), (b) the two
 * pointwise convs of the feature extractor's four AdaptiveConvNeXt1d layers (common.py:74-82), (c) the norm-FiLM projection, the two
 * coarse down convs and the mid conv, (e) the six k = 5 convs of the fused 64-channel FilterBlock.
 *   mode 1 (default since round 5, or ALIVE_DECODER_PRECISION=1): plain fp16 operands, one MFMA per product (AliveConv.precision 3,
 *          AliveGemm.planes 1), fp32 accumulate, fp32 residual streams;
 *   mode 2 (ALIVE_DECODER_PRECISION=2): two-plane split bf16 like the decoder's other GEMMs (rounds 1 - 4);
 *   mode 0: query.  Returns the mode in force.  Process-wide; not to be changed while a decoder call is in flight.
 * Measured on the reference's 450-frame fixture (tests/test_gpu_models.py::test_decoder_precision_modes): decoder waveform RMS error
 * 5.0e-6 in mode 2, 2.90e-5 in mode 1, whole conversion of a 450-frame window 1.22e-4 in both; the bar of the path is 1e-3.  Everything else in the
 * decoder (FiLM projections, input layer, to_amps, strided / transposed convs, the fused 16 / 8-channel FilterBlocks) keeps
 * split bf16 or exact fp32 in both modes, and so do both encoders (top-k / argmax downstream). */
int alive_decoder_precision(int mode);
/* Arithmetic of the pointwise convs of the ConvNeXt layers of ContentEncoder / F0Estimator on the batch path (common.py:54-62):
 *   mode 1 (default since round 5, or ALIVE_ENCODER_PRECISION=1): fp16 split planes, three MFMAs per product (AliveGemm.f16s; 22
 *          significand bits per operand);
 *   mode 2 (ALIVE_ENCODER_PRECISION=2): three bf16 planes, six MFMAs per product (24 bits; rounds 1 - 4);
 *   mode 0: query.  Both are fp32-grade: content features agree with the reference fixture to < 1e-5 of their RMS in either mode and
 * the f0 classes outside the 1e-4 margin are identical (tests/test_gpu_models.py).  In mode 1 the F0Estimator's classifier runs on
 * fp16 split planes too (its input is what last_norm leaves); the DFT and the input / output layers of both encoders stay on three
 * bf16 planes in either mode (their inputs are not range-limited by a normalisation). */
int alive_encoder_precision(int mode);
/* Values that left fp16's range (|scaled value| > 65504) while an fp16 plane was written by any kernel of modes 1 above since the last
 * clear -- they were saturated, not turned into infinities, but the result is then not the reference's.  0 on every tested checkpoint
 * and input.  SYNCHRONISES THE DEVICE (hipDeviceSynchronize, then a 4-byte read per kernel file of the CURRENT device's counters: the
 * caller's streams may be non-blocking, so the null stream alone orders nothing); reset != 0 clears the counters after the read.
 * Returns -1 when a runtime call fails -- callers must treat that as an error, not as "no saturation".
 * module/pipeline.py::Converter.convert_windows clears the counters when a batch starts (alive_f16_saturations_clear), reads them when
 * it ends and repeats the batch in modes 2 when they are not zero. */
int alive_f16_saturations(int reset);
/* Zeroes the counters asynchronously, in the order of `stream` (five 4-byte memsets; no synchronisation, graph-capturable). */
int alive_f16_saturations_clear(void* stream);

/* Decoder.forward (decoder.py:205-210) at harmonics_scale == 1:
 *   x[N][768][Lf], f0[N][Lf], phi_in[N][64] or NULL (phi = 0), crop0,
 *   wave[N][320*Lf], phi_out[N][64] at column phi_col, or NULL. */
size_t alive_decoder_workspace_bytes(int N, int Lf);
int alive_decoder_forward(const float* const* w, const float* x, const float* f0,
                          const float* phi_in, int crop0, int phi_col, int N, int Lf,
                          float* wave, float* phi_out, void* ws, void* stream);

/* pitch transform of inference.py:119-126,130 (mode 0, per-window mean pitch)
 * and realtime_inference.py:156-163 (mode 1); in place on f0[N][T]. */
int alive_pitch_transform(float* f0, int N, int T, int mode, float f0_rate, float pitch_shift,
                          float intonation, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ALIVE_VC_H */
