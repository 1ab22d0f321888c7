// Launch sequences of the three inference networks on top of the operator kernels.
//   ContentEncoder.forward     /root/reference/module/content_encoder.py:21-25
//   F0Estimator.estimate       /root/reference/module/f0_estimator.py:22-34
//   Decoder.forward            /root/reference/module/decoder.py:43-48,66-102,184-195,205-210
//   spectrogram                /root/reference/module/spectrogram.py:5-10
// Pure stream-ordered launches into caller-provided scratch: no allocation, no
// sync, so a whole forward is hipGraph-capturable.
#include <string.h>

#include <string>
#include <vector>

#include "common.h"
#include "planes_layout.h"

int alive_magnitude(const float* ri, int N, int B, int T, float* out, hipStream_t s);

namespace {

constexpr int BINS = 641, NFFT = 1280, HOP = 320;
constexpr int DFT_ROWS = 1296;                       // 2*641 = 1282 padded to x16
constexpr int CE_C = 512, CE_H = 1536, CE_OUT = 768;
constexpr int PE_C = 256, PE_H = 512, PE_OUT = 4096;
constexpr int DEC_C = 512, DEC_H = 1536, NH = 64, SEG = 320;
constexpr float SR = 16000.0f;
constexpr float NORM_EPS = 1e-4f;
constexpr int F_CH[4] = {256, 64, 16, 8};            // filter channels, coarse -> fine (decoder.py:157,171)
constexpr int F_UP[4] = {10, 8, 2, 2};               // upsampling rates, coarse -> fine
constexpr int FILM_ROWS = 6 * 2 * (256 + 64 + 16 + 8);   // 4128
// how a filter scale runs (module/_pack.py::FILTER_MODE): 0 conv by conv on the split-bf16 MFMA kernel (C = 256),
// 1 fused FilterBlock on the split-bf16 MFMA (C = 64, filter_mid.hip), 2 fused on the f32 MFMA (C = 16, 8)
constexpr int F_MODE[4] = {0, 1, 2, 2};
constexpr bool F_SPLIT[4] = {true, true, false, false};   // ConvTranspose in front of the scale on the split kernel

inline int pad16(int x) { return (x + 15) & ~15; }

// ---- weight tables --------------------------------------------------------------------------
struct Names {
    std::vector<std::string> v;
    int add(const std::string& s) { v.push_back(s); return (int)v.size() - 1; }
};

void convnext_names(Names& n, const std::string& p, bool adaptive) {
    n.add(p + ".dw_w"); n.add(p + ".dw_b");
    if (!adaptive) { n.add(p + ".norm_gain"); n.add(p + ".norm_offset"); }
    n.add(p + ".pw1.W"); n.add(p + ".pw1.b"); n.add(p + ".pw2.W"); n.add(p + ".pw2.b"); n.add(p + ".scale");
    if (!adaptive) { n.add(p + ".pw1.ws"); n.add(p + ".pw2.ws"); }      // encoders: 1 / scale of the fp16 split weight images
}

const Names& names_of(int model) {
    static Names ce, pe, dec;
    static const bool init = [&]() {                  // function-local static: initialised once, thread-safe (C++11)
        ce.add("input.W"); ce.add("input.b");
        for (int i = 0; i < 4; ++i) convnext_names(ce, "mid" + std::to_string(i), false);
        ce.add("output.W"); ce.add("output.b");
        pe.add("input.W"); pe.add("input.b");
        for (int i = 0; i < 4; ++i) convnext_names(pe, "mid" + std::to_string(i), false);
        pe.add("last_norm.gain"); pe.add("last_norm.offset"); pe.add("output.W"); pe.add("output.b"); pe.add("output.ws");
        dec.add("fe.input.W"); dec.add("fe.input.b");
        dec.add("fe.f0c1.W"); dec.add("fe.f0c1.b"); dec.add("fe.f0c2.W"); dec.add("fe.f0c2.b");
        dec.add("fe.normfilm.W"); dec.add("fe.normfilm.b");
        for (int i = 0; i < 4; ++i) convnext_names(dec, "fe.mid" + std::to_string(i), true);
        dec.add("osc.amps.W"); dec.add("osc.amps.b");
        dec.add("flt.film.W"); dec.add("flt.film.b"); dec.add("flt.film.post");
        dec.add("flt.in.W"); dec.add("flt.in.b");
        for (int i = 0; i < 4; ++i) {
            dec.add("flt.down" + std::to_string(i) + ".W"); dec.add("flt.down" + std::to_string(i) + ".b");
            if (i >= 2) dec.add("flt.down" + std::to_string(i) + ".Wp");      // the same weights as bf16 planes, tap-major (batch path)
        }
        dec.add("flt.mid.W"); dec.add("flt.mid.b");
        for (int i = 0; i < 4; ++i) {
            dec.add("flt.up" + std::to_string(i) + ".W"); dec.add("flt.up" + std::to_string(i) + ".b");
            if (F_MODE[i] == 1) { dec.add("flt.up" + std::to_string(i) + ".Wc"); dec.add("flt.up" + std::to_string(i) + ".bc"); }      // x input_conv
        }
        for (int s = 0; s < 4; ++s) {
            std::string b = "flt.blk" + std::to_string(s);
            if (F_MODE[s] == 2) { dec.add(b + ".pack"); continue; }     // fused 16- / 8-channel FilterBlock
            if (F_MODE[s] == 1) { dec.add(b + ".packW"); dec.add(b + ".packB"); }   // fused 64-channel FilterBlock (filter_mid.hip), then its
            // k5 convs one by one (filter_big.hip).  F_MODE 0: no input_conv -- module/_pack.py composes it into the transposed conv
            for (int j = 0; j < 3; ++j)
                for (int c = 1; c <= 2; ++c) {
                    std::string q = b + "." + std::to_string(j) + ".c" + std::to_string(c);
                    dec.add(q + ".W"); dec.add(q + ".b");
                }
        }
        dec.add("flt.out.W"); dec.add("flt.out.b");
        return true;
    }();
    (void)init;
    return model == 0 ? ce : (model == 1 ? pe : dec);
}

struct Table {
    const float* const* w;
    int pos;
    explicit Table(const float* const* p) : w(p), pos(0) {}
    const float* next() { return w[pos++]; }
};

// ---- conv helpers ----------------------------------------------------------------------------
AliveConv conv_desc(const float* W, const float* b, const float* X, int N, int Ci, int Tin, int Co, int KW, int stride,
                    int dil, int pad_left, int pad_mode, int Tout, float* Y) {
    AliveConv d;
    memset(&d, 0, sizeof(d));
    d.W = W; d.bias = b; d.X = X; d.N = N; d.Ci = Ci; d.Tin = Tin; d.Co = Co; d.K_pad = pad16(Ci * KW);
    d.KW = KW; d.stride = stride; d.dil = dil; d.pad_left = pad_left; d.pad_mode = pad_mode; d.Tout = Tout;
    d.up = 1; d.Y = Y;
    return d;
}
AliveConv pw_desc(const float* W, const float* b, const float* X, int N, int Ci, int T, int Co, float* Y) {
    return conv_desc(W, b, X, N, Ci, T, Co, 1, 1, 1, 0, 0, T, Y);
}

// route a conv to the split-bf16 MFMA kernel (weights packed by module/_pack.py::pack_conv_split)
AliveConv split(AliveConv d) {
    d.precision = 1;
    d.Ci_pad = (d.Ci + 31) & ~31;
    return d;
}

// ---- arithmetic of the decoder's two largest groups of GEMMs on the batch path (alive_decoder_precision) ----
// mode 1 (default since round 5): plain fp16 operands (11 significand bits; the bf16 planes carry 8), ONE MFMA per product, for
//   (a) the six k = 5 convs of the 256-channel FilterBlock (decoder.py:128-134 at the Filter's coarsest scale; AliveConv.precision 3):
//       their inputs are gelu + FiLM outputs that nothing else reads, their outputs are added to an fp32 residual stream;
//   (b) the two pointwise convs of the feature extractor's four AdaptiveConvNeXt1d layers (common.py:74-82; alive_gemm_planes with
//       planes = 1): inputs are a channel-normalised tensor and a gelu output, the output is scaled and added to the fp32 stream.
//   (c) the projection of the f0 encoding onto the AdaptiveChannelNorm scales / shifts (common.py:35-41, 512 -> 4096), the Filter's two
//       coarsest down convs (decoder.py:186-188, 64 -> 256 k8 and 256 -> 256 k10) and its mid conv (decoder.py:190);
//   (e) the six k = 5 convs of the fused 64-channel FilterBlock (filter_mid.hip, H16: a third of its MFMAs, no lo plane in its LDS image;
//       its 1x1 input conv keeps the split form).
//   Everything else -- the FiLM projections and the input layer (the waveform is 10 x more sensitive to them than to (a)), to_amps,
//   the transposed convs, the fused 16 / 8-channel FilterBlocks -- stays on two-plane split bf16 or exact fp32.  Measured on the
//   reference's 450-frame fixture: decoder waveform RMS error 5.0e-6 (mode 2) -> 2.90e-5 (mode 1; 1.40e-5 without (e)), whole
//   conversion of a 450-frame window 1.22e-4 in both modes, against the bar of 1e-3
//   (tests/test_gpu_models.py::test_decoder_precision_modes).  The first form
//   of mode 1 used plane 0 of the bf16 images (8 significand bits): 1.19e-4 on the same fixture -- fp16 costs the same MFMAs and bytes.
// mode 2: two-plane split bf16 for these too (rounds 1 - 4).  ALIVE_DECODER_PRECISION=2 or alive_decoder_precision(2).
// ALIVE_DECODER_BF16_MASK (experiments): which of (a) = 1, (b) = 2, (c) = 4, (d: the two transposed convs) = 8, (e) = 16 mode 1 covers
// (default 23).
// Sensitivities (oracle with the operands of one group rounded to fp16 | to bf16; 450-frame fixture, waveform RMS 0.66): (a) 1.24e-5 |
// 1.11e-4, (b) 3.4e-6 | 2.7e-5, (c) 1.3e-6 + 1.2e-6 + 2.0e-7 | 2.2e-5 + 8.3e-6 + 1.3e-6; not adopted: FiLM projections 1.4e-4 | 1.6e-3,
// input layer 1.2e-4 | 1.5e-3, up convs 1.0e-5 | 8.1e-5 (store bound: nothing to gain); (e) 2.6e-5 | 2.2e-4 -- adopted last, for
// 3.7 ms per step (the block is vector-issue bound, but a k-step with one MFMA instead of three is 30 % shorter).
int g_decoder_precision = 0;      // 0: not decided yet (environment, else 1)
int decoder_precision() {
    if (g_decoder_precision == 0) {
        const char* e = getenv("ALIVE_DECODER_PRECISION");
        g_decoder_precision = (e != nullptr && atoi(e) == 2) ? 2 : 1;
    }
    return g_decoder_precision;
}
int decoder_bf16_mask() {
    static const int m = getenv("ALIVE_DECODER_BF16_MASK") ? atoi(getenv("ALIVE_DECODER_BF16_MASK")) : 23;
    return decoder_precision() == 1 ? m : 0;
}

bool fb64s_enabled() {          // ALIVE_FB64S=0: the 64-channel block on filter_mid.hip's sweep kernel, as before
    static const bool on = !(getenv("ALIVE_FB64S") && atoi(getenv("ALIVE_FB64S")) == 0);
    return on;
}
bool fb256_enabled() {          // ALIVE_FB256=0: the 256-channel block conv by conv, as in round 5
    static const bool on = !(getenv("ALIVE_FB256") && atoi(getenv("ALIVE_FB256")) == 0);
    return on;
}

// the plain (fp16) image of a weight tensor packed by module/_pack.py::pack_conv_split_h: the third slab behind the two bf16 planes
const float* plain_w(const float* W, int rows_pad16, int K) {
    return (const float*)((const unsigned short*)W + (size_t)2 * rows_pad16 * K);
}
AliveConv plain(AliveConv d) {           // d: a split() descriptor (precision 1, W = the two-plane pack + fp16 slab)
    d.precision = 3;
    d.W = plain_w(d.W, (d.Co + 15) & ~15, d.KW * d.Ci_pad);
    return d;
}

AliveConv split3(AliveConv d) {      // 3-plane split ("bf16x6"): fp32-grade, used by the encoders (argmax / top-k downstream)
    d = split(d);
    d.precision = 2;
    return d;
}

#define RUN(expr)                     \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != ALIVE_OK) return rc_; \
    } while (0)

struct ConvNeXtW {
    const float *dw_w, *dw_b, *gain, *offset, *pw1W, *pw1b, *pw2W, *pw2b, *scale, *ws1, *ws2;
    ConvNeXtW(Table& t, bool adaptive) {
        dw_w = t.next(); dw_b = t.next();
        gain = offset = ws1 = ws2 = nullptr;
        if (!adaptive) { gain = t.next(); offset = t.next(); }
        pw1W = t.next(); pw1b = t.next(); pw2W = t.next(); pw2b = t.next(); scale = t.next();
        if (!adaptive) { ws1 = t.next(); ws2 = t.next(); }
    }
};

// ---- frame-rate 1x1 convs ---------------------------------------------------------------------
// With enough columns (N*T) they run as plane-packed GEMMs (gemm_planes.hip): the activation operand is split into
// bf16 planes once, by its producer or by alive_to_planes, and both operands stream into LDS by LDS-DMA.  Below that
// (the streaming path: a handful of frames per step) the fp32-activation kernels of alive_conv1d stay in charge.
constexpr int64_t PLANES_MIN_COLS = 96;
inline bool use_planes(int N, int T) { return (int64_t)N * T >= PLANES_MIN_COLS; }
inline size_t planes_bytes(int N, int T, int C, int planes) { return align_up(alive_planes_bytes((int64_t)N * T, C, planes), 256); }

int pw_gemm(const float* W, const float* b, const void* P, int N, int T, int Ci, int Co, int planes, int act,
            const float* post_add, const float* ch_scale, const float* residual, float* Y, void* Pout, void* s) {
    AliveGemm g;
    memset(&g, 0, sizeof(g));
    g.W = W; g.bias = b; g.P = P; g.N = N; g.T = T; g.Ci = Ci; g.Co = Co; g.planes = planes; g.act = act;
    g.post_add = post_add; g.ch_scale = ch_scale; g.residual = residual; g.Y = Y; g.Pout = Pout;
    return alive_gemm_planes(&g, s);
}

// x <- x + scale * pw2(gelu(pw1(norm(dw(x)))))      (common.py:54-62 / 74-82)
// Pa / Ph: plane-packed scratch for the normalised input and the hidden layer (nullptr: fp32-activation path via hbuf)
// ---- arithmetic of the encoders' ConvNeXt pointwise convs on the batch path (alive_encoder_precision) ----
// mode 1 (default since round 5): fp16 SPLIT planes -- hi = fp16(s v), lo = fp16(s v - hi) of a power-of-two multiple of the values, three
//   MFMAs per product (hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_f16), 22 significand bits per operand -- instead of three bf16 planes
//   and six MFMAs (24 bits).  The operands are what a ChannelNorm and a gelu leave: bounded, so a fixed activation scale of 2^8 keeps
//   them in fp16's range (saturation at +-65504 = |v| > 255 would need a norm gain above 11); elements below 2^-11 / s lose relative,
//   not absolute, precision (error < 2^-33).  Weights carry a per-tensor scale (module/_pack.py::pack_conv_split_f16s).
// mode 2: the three-plane bf16 form (rounds 1 - 4).  ALIVE_ENCODER_PRECISION=2 or alive_encoder_precision(2).
// The DFT, the input / output layers and the classifier stay on three bf16 planes in both modes: their inputs (waveform, magnitudes)
// have no normalisation in front and bf16 planes keep fp32's range.
int g_encoder_precision = 0;
int encoder_precision() {
    if (g_encoder_precision == 0) {
        const char* e = getenv("ALIVE_ENCODER_PRECISION");
        g_encoder_precision = (e != nullptr && atoi(e) == 2) ? 2 : 1;
    }
    return g_encoder_precision;
}
constexpr float F16S_ACT_SCALE = 256.0f;      // activations of the fp16 split form: 2^8

int convnext_layer(const ConvNeXtW& w, float* x, float* ybuf, float* hbuf, void* Pa, void* Ph, int N, int C, int H, int T,
                   const float* cond, int cond_rows, int scale_row, int shift_row, int planes, void* s) {
    if (Pa != nullptr && planes == 3 && w.ws1 != nullptr && encoder_precision() == 1) {
        RUN(alive_dwconv_norm_planes_f16s(x, N, C, T, w.dw_w, w.dw_b, cond ? 1 : 0, w.gain, w.offset, cond, cond_rows, scale_row,
                                          shift_row, NORM_EPS, F16S_ACT_SCALE, Pa, s));
        AliveGemm g;
        memset(&g, 0, sizeof(g));
        // (the fp16 pair sits behind the three bf16 planes of the same tensor)
        g.W = (const float*)((const unsigned short*)w.pw1W + (size_t)3 * ((H + 15) & ~15) * ((C + 31) & ~31));
        g.bias = w.pw1b; g.P = Pa; g.N = N; g.T = T; g.Ci = C; g.Co = H; g.planes = 2; g.act = 1; g.Pout = Ph;
        g.f16s = 1; g.wscale = w.ws1; g.in_unscale = 1.0f / F16S_ACT_SCALE; g.pout_scale = F16S_ACT_SCALE;
        RUN(alive_gemm_planes(&g, s));
        memset(&g, 0, sizeof(g));
        g.W = (const float*)((const unsigned short*)w.pw2W + (size_t)3 * ((C + 15) & ~15) * ((H + 31) & ~31));
        g.bias = w.pw2b; g.P = Ph; g.N = N; g.T = T; g.Ci = H; g.Co = C; g.planes = 2; g.act = 0;
        g.ch_scale = w.scale; g.residual = x; g.Y = x;
        g.f16s = 1; g.wscale = w.ws2; g.in_unscale = 1.0f / F16S_ACT_SCALE;
        return alive_gemm_planes(&g, s);
    }
    if (Pa != nullptr) {
        const int gp = (planes == 2 && (decoder_bf16_mask() & 2)) ? 1 : planes;     // the decoder's layers (the encoders run three planes)
        RUN(alive_dwconv_norm_planes(x, N, C, T, w.dw_w, w.dw_b, cond ? 1 : 0, w.gain, w.offset, cond, cond_rows, scale_row,
                                     shift_row, NORM_EPS, gp, Pa, s));
        const float* W1 = gp == 1 ? plain_w(w.pw1W, (H + 15) & ~15, (C + 31) & ~31) : w.pw1W;
        const float* W2 = gp == 1 ? plain_w(w.pw2W, (C + 15) & ~15, (H + 31) & ~31) : w.pw2W;
        RUN(pw_gemm(W1, w.pw1b, Pa, N, T, C, H, gp, 1, nullptr, nullptr, nullptr, nullptr, Ph, s));
        return pw_gemm(W2, w.pw2b, Ph, N, T, H, C, gp, 0, nullptr, w.scale, x, x, nullptr, s);
    }
    RUN(alive_dwconv_norm(x, N, C, T, w.dw_w, w.dw_b, cond ? 1 : 0, w.gain, w.offset, cond, cond_rows, scale_row,
                          shift_row, NORM_EPS, ybuf, s));
    AliveConv d1 = pw_desc(w.pw1W, w.pw1b, ybuf, N, C, T, H, hbuf);
    d1.act = 1;
    d1 = planes == 3 ? split3(d1) : split(d1);
    RUN(alive_conv1d(&d1, s));
    AliveConv d2 = pw_desc(w.pw2W, w.pw2b, hbuf, N, H, T, C, x);
    d2.ch_scale = w.scale;
    d2.residual = x;
    d2 = planes == 3 ? split3(d2) : split(d2);
    RUN(alive_conv1d(&d2, s));
    return ALIVE_OK;
}

// one 1x1 conv from an fp32 [N][Ci][T] tensor: converts to planes (scratch Pa) when the plane path is on
int pw_conv(const float* W, const float* b, const float* X, void* Pa, int N, int T, int Ci, int Co, int planes, int act,
            const float* post_add, float* Y, void* s) {
    if (Pa != nullptr) {
        RUN(alive_to_planes(X, N, Ci, T, planes, Pa, s));
        return pw_gemm(W, b, Pa, N, T, Ci, Co, planes, act, post_add, nullptr, nullptr, Y, nullptr, s);
    }
    AliveConv d = pw_desc(W, b, X, N, Ci, T, Co, Y);
    d.act = act;
    d.post_add = post_add;
    d = planes == 3 ? split3(d) : split(d);
    return alive_conv1d(&d, s);
}

__global__ void dft_basis_kernel(float* basis) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= DFT_ROWS * NFFT) return;
    int row = i / NFFT, j = i % NFFT;
    float v = 0.0f;
    if (row < 2 * BINS) {
        int f = row < BINS ? row : row - BINS;
        int ph = (int)(((long long)f * j) % NFFT);
        double sn, cs;
        sincospi(2.0 * (double)ph / (double)NFFT, &sn, &cs);
        v = row < BINS ? (float)cs : (float)(-sn);
    }
    basis[i] = v;
}

}  // namespace

extern "C" int alive_weight_count(int model) { return (model >= 0 && model <= 2) ? (int)names_of(model).v.size() : -1; }
extern "C" const char* alive_weight_name(int model, int index) {
    if (model < 0 || model > 2) return nullptr;
    const Names& n = names_of(model);
    return (index >= 0 && index < (int)n.v.size()) ? n.v[index].c_str() : nullptr;
}

// ---- spectrogram -------------------------------------------------------------------------------
// basis buffer: fp32 [1296][1280] (streaming path: exact f32-MFMA conv), then bf16 3 planes x 1296 rows x 1280, k-blocked (batch path)
int alive_gelu_film_impl(const float* H, int N, int C, int L, const float* film, int film_rows, int Lf, int scale_row,
                         int shift_row, int t0, int f0, int film_ld, float* Z, void* Zp, int planes, void* stream);
namespace {
constexpr size_t BASIS_F32 = (size_t)DFT_ROWS * NFFT;

__global__ void dft_basis_planes_kernel(unsigned short* planes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= DFT_ROWS * NFFT) return;
    int row = i / NFFT, j = i % NFFT;
    float v = 0.0f;
    if (row < 2 * BINS) {
        int f = row < BINS ? row : row - BINS;
        int ph = (int)(((long long)f * j) % NFFT);
        double sn, cs;
        sincospi(2.0 * (double)ph / (double)NFFT, &sn, &cs);
        v = row < BINS ? (float)cs : (float)(-sn);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const unsigned short h = f32_to_bf16_rn(v);
        planes[planes_at(pl, row, j, DFT_ROWS, NFFT)] = h;              // the GEMM's weight operand: k-blocked (planes_layout.h)
        v -= __uint_as_float((unsigned)h << 16);
    }
}

// the same basis with (re, im) of a bin in ADJACENT rows (row 2f = cos, 2f + 1 = -sin): the weight operand of the fused front end,
// whose GEMM epilogue takes the magnitude of each row pair (AliveGemm.act == 4)
__global__ void dft_basis_planes_il_kernel(unsigned short* planes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= DFT_ROWS * NFFT) return;
    int row = i / NFFT, j = i % NFFT;
    float v = 0.0f;
    if (row < 2 * BINS) {
        int f = row >> 1;
        int ph = (int)(((long long)f * j) % NFFT);
        double sn, cs;
        sincospi(2.0 * (double)ph / (double)NFFT, &sn, &cs);
        v = (row & 1) == 0 ? (float)cs : (float)(-sn);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const unsigned short h = f32_to_bf16_rn(v);
        planes[planes_at(pl, row, j, DFT_ROWS, NFFT)] = h;
        v -= __uint_as_float((unsigned)h << 16);
    }
}

// wav[N][L] -> planes [3][N][L + 1280] of the centre-padded signal (reflect, torch.stft center=True)
__global__ void wav_to_planes_kernel(const float* __restrict__ wav, int L, int Lpad, size_t plane_stride, unsigned short* __restrict__ P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Lpad) return;
    const int n = blockIdx.y;
    int sidx = i - NFFT / 2;
    sidx = sidx < 0 ? -sidx : sidx;
    sidx = sidx >= L ? 2 * (L - 1) - sidx : sidx;
    float v = wav[(size_t)n * L + sidx];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const unsigned short h = f32_to_bf16_rn(v);
        P[(size_t)pl * plane_stride + (size_t)n * Lpad + i] = h;
        v -= __uint_as_float((unsigned)h << 16);
    }
}
}  // namespace

// basis buffer: fp32 image, bf16 planes (rows re | im), bf16 planes (rows interleaved: the fused front end)
extern "C" size_t alive_dft_basis_bytes(void) { return BASIS_F32 * sizeof(float) + 2 * 3 * BASIS_F32 * sizeof(unsigned short); }
extern "C" int alive_dft_basis(float* basis, void* stream) {
    ALIVE_CHECK_ARG(basis, "alive_dft_basis: null");
    dft_basis_kernel<<<cdiv((int64_t)DFT_ROWS * NFFT, 256), 256, 0, (hipStream_t)stream>>>(basis);
    dft_basis_planes_kernel<<<cdiv((int64_t)DFT_ROWS * NFFT, 256), 256, 0, (hipStream_t)stream>>>((unsigned short*)(basis + BASIS_F32));
    dft_basis_planes_il_kernel<<<cdiv((int64_t)DFT_ROWS * NFFT, 256), 256, 0, (hipStream_t)stream>>>((unsigned short*)(basis + BASIS_F32) + 3 * BASIS_F32);
    ALIVE_CHECK_LAUNCH("alive_dft_basis");
    return ALIVE_OK;
}
extern "C" size_t alive_spectrogram_workspace_bytes(int N, int L) {
    return align_up((size_t)N * 2 * BINS * (L / HOP) * sizeof(float), 256) + align_up((size_t)3 * N * (L + NFFT) * 2, 256) + 512;
}
extern "C" int alive_spectrogram(const float* basis, const float* wav, int N, int L, float* spec, void* ws, void* stream) {
    ALIVE_CHECK_ARG(basis && wav && spec && ws, "alive_spectrogram: null pointer");
    ALIVE_CHECK_ARG(N > 0 && L > NFFT / 2 && L >= HOP, "alive_spectrogram: needs L > %d samples (reflect pad), got %d", NFFT / 2, L);
    const int T = L / HOP;
    Arena a(ws);
    float* ri = a.take<float>((size_t)N * 2 * BINS * T);
    if (use_planes(N, T) && L % 8 == 0) {
        // batch path: the padded signal is split into three bf16 planes ONCE; the frames are overlapping rows of it
        // (row stride = hop), so the DFT is a plane-packed GEMM ("bf16x6", fp32-grade) without an im2col
        const int Lpad = L + NFFT;
        unsigned short* wp = a.take<unsigned short>((size_t)3 * N * Lpad);
        wav_to_planes_kernel<<<dim3(cdiv(Lpad, 256), N), 256, 0, (hipStream_t)stream>>>(wav, L, Lpad, (size_t)N * Lpad, wp);
        AliveGemm g;
        memset(&g, 0, sizeof(g));
        g.W = basis + BASIS_F32; g.P = wp; g.N = N; g.T = T; g.Ci = NFFT; g.Co = 2 * BINS; g.planes = 3; g.Y = ri;
        g.b_plane = (int64_t)N * Lpad; g.b_win = Lpad; g.b_row = HOP;
        RUN(alive_gemm_planes(&g, stream));
        return alive_magnitude(ri, N, BINS, T, spec, (hipStream_t)stream);
    }
    AliveConv d = conv_desc(basis, nullptr, wav, N, 1, L, 2 * BINS, NFFT, HOP, 1, NFFT / 2, 2, T, ri);
    RUN(alive_conv1d(&d, stream));
    return alive_magnitude(ri, N, BINS, T, spec, (hipStream_t)stream);
}

// ---- content encoder -------------------------------------------------------------------------
namespace {
struct EncBuffers {
    float *x, *y, *h, *lg;
    void *Pa, *Ph;
    size_t bytes;
};
// C / H: ConvNeXt widths; logits: rows of the fp32 logits tensor of the f0 estimator (0 for the content encoder)
EncBuffers enc_layout(void* ws, int N, int T, int C, int H, int logits) {
    Arena a(ws);
    EncBuffers b;
    const size_t f = (size_t)N * T;
    b.x = a.take<float>(f * C);
    b.y = a.take<float>(f * C);
    b.lg = logits ? a.take<float>(f * logits) : nullptr;
    if (use_planes(N, T)) {
        b.h = nullptr;
        b.Pa = a.take<char>(planes_bytes(N, T, BINS, 3));          // widest plane-packed input: the 641 spectrogram bins
        b.Ph = a.take<char>(planes_bytes(N, T, H, 3));
    } else {
        b.h = a.take<float>(f * H);
        b.Pa = b.Ph = nullptr;
    }
    b.bytes = a.used() + 1024;
    return b;
}
}  // namespace

namespace {
// the networks behind their input layers: b.x holds input_layer(spec) (content_encoder.py:22 / f0_estimator.py:23), t stands behind
// the input layer's two table entries
int ce_body(Table& t, EncBuffers& b, int N, int T, float* out, void* stream);
int pe_body(Table& t, EncBuffers& b, int N, int T, float* f0, void* stream);
}  // namespace

extern "C" size_t alive_content_encoder_workspace_bytes(int N, int T) { return enc_layout(nullptr, N, T, CE_C, CE_H, 0).bytes; }
extern "C" int alive_content_encoder(const float* const* w, const float* spec, int N, int T, float* out, void* ws, void* stream) {
    ALIVE_CHECK_ARG(w && spec && out && ws && N > 0 && T > 0, "alive_content_encoder: bad args");
    Table t(w);
    EncBuffers b = enc_layout(ws, N, T, CE_C, CE_H, 0);
    const float* inW = t.next(); const float* inb = t.next();
    RUN(pw_conv(inW, inb, spec, b.Pa, N, T, BINS, CE_C, 3, 0, nullptr, b.x, stream));
    return ce_body(t, b, N, T, out, stream);
}
namespace {
int ce_body(Table& t, EncBuffers& b, int N, int T, float* out, void* stream) {
    for (int i = 0; i < 4; ++i) {
        ConvNeXtW cw(t, false);
        RUN(convnext_layer(cw, b.x, b.y, b.h, b.Pa, b.Ph, N, CE_C, CE_H, T, nullptr, 0, 0, 0, 3, stream));
    }
    const float* oW = t.next(); const float* ob = t.next();
    return pw_conv(oW, ob, b.x, b.Pa, N, T, CE_C, CE_OUT, 3, 0, nullptr, out, stream);
}
}  // namespace

// ---- f0 estimator ----------------------------------------------------------------------------
extern "C" size_t alive_f0_estimate_workspace_bytes(int N, int T) { return enc_layout(nullptr, N, T, PE_C, PE_H, PE_OUT).bytes; }
extern "C" int alive_f0_estimate(const float* const* w, const float* spec, int N, int T, float* f0, void* ws, void* stream) {
    ALIVE_CHECK_ARG(w && spec && f0 && ws && N > 0 && T > 0, "alive_f0_estimate: bad args");
    Table t(w);
    EncBuffers b = enc_layout(ws, N, T, PE_C, PE_H, PE_OUT);
    const float* inW = t.next(); const float* inb = t.next();
    RUN(pw_conv(inW, inb, spec, b.Pa, N, T, BINS, PE_C, 3, 0, nullptr, b.x, stream));
    return pe_body(t, b, N, T, f0, stream);
}
namespace {
int pe_body(Table& t, EncBuffers& b, int N, int T, float* f0, void* stream) {
    for (int i = 0; i < 4; ++i) {
        ConvNeXtW cw(t, false);
        RUN(convnext_layer(cw, b.x, b.y, b.h, b.Pa, b.Ph, N, PE_C, PE_H, T, nullptr, 0, 0, 0, 3, stream));
    }
    const float* g = t.next(); const float* of = t.next();
    const float* oW = t.next(); const float* ob = t.next(); const float* ows = t.next();
    if (b.Pa != nullptr && encoder_precision() == 1) {
        // the classifier on fp16 split planes (its input is what last_norm leaves: bounded), argmax in the epilogue as below
        RUN(alive_dwconv_norm_planes_f16s(b.x, N, PE_C, T, nullptr, nullptr, 0, g, of, nullptr, 0, 0, 0, NORM_EPS, F16S_ACT_SCALE, b.Pa, stream));
        const int64_t cols = (int64_t)N * T;
        const int nblk = (PE_OUT + 63) / 64;
        AliveGemm gm;
        memset(&gm, 0, sizeof(gm));
        gm.W = (const float*)((const unsigned short*)oW + (size_t)3 * ((PE_OUT + 15) & ~15) * ((PE_C + 31) & ~31));
        gm.bias = ob; gm.P = b.Pa; gm.N = N; gm.T = T; gm.Ci = PE_C; gm.Co = PE_OUT; gm.planes = 2; gm.act = 3;
        gm.f16s = 1; gm.wscale = ows; gm.in_unscale = 1.0f / F16S_ACT_SCALE;
        gm.arg_val = b.lg;
        gm.arg_idx = (int32_t*)(b.lg + (size_t)nblk * cols);
        RUN(alive_gemm_planes(&gm, stream));
        return alive_argmax_merge(gm.arg_val, gm.arg_idx, nblk, cols, f0, stream);
    }
    if (b.Pa != nullptr) {
        RUN(alive_dwconv_norm_planes(b.x, N, PE_C, T, nullptr, nullptr, 0, g, of, nullptr, 0, 0, 0, NORM_EPS, 3, b.Pa, stream));
        // 256 -> 4096 classes with the argmax in the GEMM's epilogue: the logits (16 KB per frame) are never stored;
        // b.lg holds the per-64-row-block candidates instead (512 B per frame)
        const int nblk = (PE_OUT + 63) / 64;
        const int64_t cols = (int64_t)N * T;
        AliveGemm g;
        memset(&g, 0, sizeof(g));
        g.W = oW; g.bias = ob; g.P = b.Pa; g.N = N; g.T = T; g.Ci = PE_C; g.Co = PE_OUT; g.planes = 3; g.act = 3;
        g.arg_val = b.lg;
        g.arg_idx = (int32_t*)(b.lg + (size_t)nblk * cols);
        RUN(alive_gemm_planes(&g, stream));
        return alive_argmax_merge(g.arg_val, g.arg_idx, nblk, cols, f0, stream);
    } else {
        RUN(alive_channel_norm(b.x, N, PE_C, T, g, of, NORM_EPS, b.y, stream));
        RUN(pw_conv(oW, ob, b.y, b.Pa, N, T, PE_C, PE_OUT, 3, 0, nullptr, b.lg, stream));
    }
    return alive_argmax_channels(b.lg, N, PE_OUT, T, f0, stream);
}
}  // namespace

// ---- fused front end: waveform -> (content features, f0 classes) ------------------------------------------
// spectrogram.py:5-10 + content_encoder.py:21-25 + f0_estimator.py:22-34 without an fp32 spectrogram: the DFT GEMM's epilogue takes
// the magnitudes and writes them as the 3-plane k-blocked image both input layers read (AliveGemm.act == 4), and those two layers
// run as ONE 641 -> 512 + 256 GEMM over it (AliveGemm.y_split).  Against alive_spectrogram + alive_f0_estimate +
// alive_content_encoder: the [N][1282][T] re / im tensor, the [N][641][T] magnitudes and two alive_to_planes passes are gone
// (1.5 GB of HBM traffic per 128 windows), every value is bitwise the same.  Batch path only (>= 96 frame columns, L % 8 == 0).
//   w_in / b_in: the two input layers' packed weights concatenated along the rows ([3][672 / 32][768][32] bf16 planes: CE rows, then
//   PE rows) and their biases [768] -- module/ops.py::front_end builds them once per pair of networks.
namespace {
struct FrontBuffers {
    unsigned short* wp;      // padded signal planes
    char* Ps;                // magnitude planes [3][672 / 32][cols_pad][32]
    void *ce_ws, *pe_ws;
    size_t bytes;
};
FrontBuffers front_layout(void* ws, int N, int L) {
    Arena a(ws);
    FrontBuffers b;
    const int T = L / HOP;
    b.wp = a.take<unsigned short>((size_t)3 * N * (L + NFFT));
    b.Ps = a.take<char>(planes_bytes(N, T, BINS, 3));
    b.ce_ws = a.take<char>(enc_layout(nullptr, N, T, CE_C, CE_H, 0).bytes);
    b.pe_ws = a.take<char>(enc_layout(nullptr, N, T, PE_C, PE_H, PE_OUT).bytes);
    b.bytes = a.used() + 1024;
    return b;
}
}  // namespace
extern "C" size_t alive_front_end_workspace_bytes(int N, int L) { return front_layout(nullptr, N, L).bytes; }
extern "C" int alive_front_end(const float* basis, const float* const* ce_w, const float* const* pe_w, const void* w_in, const float* b_in,
                               const float* wav, int N, int L, float* feat, float* f0, void* ws, void* stream) {
    ALIVE_CHECK_ARG(basis && ce_w && pe_w && w_in && b_in && wav && feat && f0 && ws, "alive_front_end: null pointer");
    ALIVE_CHECK_ARG(N > 0 && L > NFFT / 2 && L >= HOP, "alive_front_end: needs L > %d samples (reflect pad), got %d", NFFT / 2, L);
    const int T = L / HOP;
    ALIVE_CHECK_ARG(use_planes(N, T) && L % 8 == 0, "alive_front_end: batch path only (N * T >= %d frame columns, L %% 8 == 0): use alive_spectrogram + "
                    "alive_f0_estimate + alive_content_encoder", (int)PLANES_MIN_COLS);
    FrontBuffers fb = front_layout(ws, N, L);
    const int Lpad = L + NFFT;
    wav_to_planes_kernel<<<dim3(cdiv(Lpad, 256), N), 256, 0, (hipStream_t)stream>>>(wav, L, Lpad, (size_t)N * Lpad, fb.wp);
    AliveGemm g;
    memset(&g, 0, sizeof(g));
    g.W = (const unsigned short*)(basis + BASIS_F32) + 3 * BASIS_F32; g.P = fb.wp; g.N = N; g.T = T; g.Ci = NFFT; g.Co = 2 * BINS; g.planes = 3;
    g.act = 4; g.Pout = fb.Ps;
    g.b_plane = (int64_t)N * Lpad; g.b_win = Lpad; g.b_row = HOP;
    RUN(alive_gemm_planes(&g, stream));
    EncBuffers ce = enc_layout(fb.ce_ws, N, T, CE_C, CE_H, 0), pe = enc_layout(fb.pe_ws, N, T, PE_C, PE_H, PE_OUT);
    memset(&g, 0, sizeof(g));
    g.W = w_in; g.bias = b_in; g.P = fb.Ps; g.N = N; g.T = T; g.Ci = BINS; g.Co = CE_C + PE_C; g.planes = 3;
    g.Y = ce.x; g.Y2 = pe.x; g.y_split = CE_C;
    RUN(alive_gemm_planes(&g, stream));
    Table tc(ce_w), tp(pe_w);
    tc.next(); tc.next(); tp.next(); tp.next();          // the input layers ran above
    RUN(pe_body(tp, pe, N, T, f0, stream));
    return ce_body(tc, ce, N, T, feat, stream);
}

// ---- decoder ---------------------------------------------------------------------------------
namespace {
struct DecBuffers {
    float *x, *y, *h, *sinb, *cond, *normfilm, *amps, *src, *film, *d0, *d1, *d2, *d3, *m, *U, *Hh, *Zz, *Z2;
    void *Pa, *Ph;          // plane-packed scratch of the FeatureExtractor GEMMs (nullptr on the streaming path)
    void* osc_ws;
    size_t bytes;
};
DecBuffers dec_layout(void* ws, int N, int Lf) {
    Arena a(ws);
    DecBuffers b;
    const size_t f = (size_t)N * Lf, Lw = (size_t)Lf * SEG;
    b.x = a.take<float>(f * DEC_C);
    b.y = a.take<float>(f * DEC_C);
    if (use_planes(N, Lf)) {
        b.h = nullptr;
        b.Pa = a.take<char>(planes_bytes(N, Lf, 768, 2));
        b.Ph = a.take<char>(planes_bytes(N, Lf, DEC_H, 2));
    } else {
        b.h = a.take<float>(f * DEC_H);
        b.Pa = b.Ph = nullptr;
    }
    b.sinb = a.take<float>(f * DEC_C);
    b.cond = a.take<float>(f * DEC_C);
    b.normfilm = a.take<float>(f * 4096);
    b.amps = a.take<float>(f * NH);
    b.src = a.take<float>((size_t)N * Lw);
    b.film = a.take<float>(f * FILM_ROWS);
    b.d0 = a.take<float>((size_t)N * 16 * (Lw / 2));
    b.d1 = a.take<float>((size_t)N * 64 * (Lw / 4));
    b.d2 = a.take<float>((size_t)N * 256 * (Lw / 32));
    b.d3 = a.take<float>(f * 256);
    b.m = a.take<float>(f * 256);
    const size_t big = (size_t)N * 64 * (Lw / 4);      // largest filter-scale tensor (64 x 36000 per window)
    b.U = a.take<float>(big + 32768);                  // + the column padding of the planes of d1 it doubles as scratch for
    b.Hh = a.take<float>(big);
    b.Zz = a.take<float>(big + 32768);                 // (planes of d2)
    b.Z2 = a.take<float>(big + 32768);                 // (+ the 128-column padding of a plane-packed image)
    b.osc_ws = a.take<char>(alive_oscillator_workspace_bytes(N, NH, Lf));
    b.bytes = a.used() + 1024;
    return b;
}
}  // namespace

extern "C" size_t alive_decoder_workspace_bytes(int N, int Lf) { return dec_layout(nullptr, N, Lf).bytes; }

namespace {
__global__ void slice_frames_kernel(const float* __restrict__ f0, int Lf, int f_begin, int n_frames, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_frames) out[(size_t)blockIdx.y * n_frames + i] = f0[(size_t)blockIdx.y * Lf + f_begin + i];
}

// Decoder.forward on the frames [fb, fb + Lf) of windows of Lw_frames frames (fb = 0, Lf = Lw_frames: the whole window).
// f0_win: f0 of the whole window (oscillator phase); f0: the range's own frames, contiguous [N][Lf].
int decoder_run(const float* const* w, const float* x_in, const float* f0, const float* f0_win, const float* phi_in, int crop0,
                int phi_col, int N, int Lf, int Lw_frames, int f_begin, float* wave, float* phi_out, DecBuffers& b, void* stream) {
    const bool ranged = Lf != Lw_frames;
    Table t(w);
    const int Lw = Lf * SEG;
    // -- FeatureExtractor (decoder.py:43-48)
    const float* inW = t.next(); const float* inb = t.next();
    RUN(pw_conv(inW, inb, x_in, b.Pa, N, Lf, 768, DEC_C, 2, 0, nullptr, b.x, stream));
    const float* c1W = t.next(); const float* c1b = t.next(); const float* c2W = t.next(); const float* c2b = t.next();
    { AliveConv d = pw_desc(c1W, c1b, f0, N, 1, Lf, DEC_C, b.sinb); d.act = 3; RUN(alive_conv1d(&d, stream)); }
    RUN(pw_conv(c2W, c2b, b.sinb, b.Pa, N, Lf, DEC_C, DEC_C, 2, 0, nullptr, b.cond, stream));
    const float* nfW = t.next(); const float* nfb = t.next();
    if (b.Pa != nullptr && (decoder_bf16_mask() & 4)) {      // (c) plain fp16
        RUN(alive_to_planes(b.cond, N, DEC_C, Lf, 1, b.Pa, stream));
        RUN(pw_gemm(plain_w(nfW, 4096, DEC_C), nfb, b.Pa, N, Lf, DEC_C, 4096, 1, 0, nullptr, nullptr, nullptr, b.normfilm, nullptr, stream));
    } else {
        RUN(pw_conv(nfW, nfb, b.cond, b.Pa, N, Lf, DEC_C, 4096, 2, 0, nullptr, b.normfilm, stream));
    }
    for (int i = 0; i < 4; ++i) {
        ConvNeXtW cw(t, true);
        RUN(convnext_layer(cw, b.x, b.y, b.h, b.Pa, b.Ph, N, DEC_C, DEC_H, Lf, b.normfilm, 4096, i * 1024, i * 1024 + 512, 2, stream));
    }
    // -- HarmonicOscillator (decoder.py:66-102)
    const float* aW = t.next(); const float* ab = t.next();
    RUN(pw_conv(aW, ab, b.x, b.Pa, N, Lf, DEC_C, NH, 2, 2, nullptr, b.amps, stream));
    RUN(alive_oscillator_range(b.amps, f0_win, phi_in, N, NH, Lw_frames, SEG, SR, crop0, phi_col, f_begin, Lf, b.src, phi_out, b.osc_ws, stream));
    // -- Filter (decoder.py:184-195); FiLM scale(+1)/shift of all 24 modulated convs in one GEMM
    const float* fW = t.next(); const float* fb = t.next(); const float* fpost = t.next();
    if (b.Pa != nullptr) {      // b.Pa still holds the planes of b.x (to_amps above)
        RUN(pw_gemm(fW, fb, b.Pa, N, Lf, DEC_C, FILM_ROWS, 2, 0, fpost, nullptr, nullptr, b.film, nullptr, stream));
    } else {
        AliveConv d = split(pw_desc(fW, fb, b.x, N, DEC_C, Lf, FILM_ROWS, b.film)); d.post_add = fpost; RUN(alive_conv1d(&d, stream));
    }
    const float* siW = t.next(); const float* sib = t.next();
    const int dch[5] = {8, 16, 64, 256, 256};
    const int drate[4] = {2, 2, 8, 10};
    float* dbuf[5] = {nullptr, b.d0, b.d1, b.d2, b.d3};
    {   // source_in + downs[0] in one streaming kernel: the 8-channel tensor in between is not a skip
        const float* W = t.next(); const float* bb = t.next();
        RUN(alive_filter_source_in(b.src, N, Lw, siW, sib, W, bb, b.d0, stream));
    }
    int len = Lw / 2;
    for (int i = 1; i < 4; ++i) {
        const float* W = t.next(); const float* bb = t.next();
        const float* Wp = i >= 2 ? t.next() : nullptr;
        const int r = drate[i];
        if (Wp != nullptr && b.Pa != nullptr) {
            // Conv1d(k == stride == r, no padding) on the batch path: with the input as k-blocked bf16 planes, the K vector of
            // output column t is the r consecutive plane rows r t .. r t + r - 1 in each of the cpad / 32 k-blocks -- a plane GEMM
            // whose B rows overlap nothing and start every r rows (custom row placement, AliveGemm.b_cblk), 16/3 the rate of the
            // exact-fp32 kernel
            void* Pd = i == 2 ? (void*)b.U : (void*)b.Zz;             // both idle until the up path starts
            const int cpad = (dch[i] + 31) & ~31;
            // (i == 2: the planes of d1 were written by downs[1] beside the fp32 skip tensor, AliveConv.Yp; i == 3: left there by downs[2]'s epilogue)
            AliveGemm g;
            memset(&g, 0, sizeof(g));
            g.W = Wp; g.bias = bb; g.P = Pd; g.N = N; g.T = len / r; g.Ci = r * cpad; g.Co = dch[i + 1];
            g.planes = (decoder_bf16_mask() & 4) ? 1 : 2;            // (c) plain fp16: d1 / d2 arrive as one fp16 plane
            if (g.planes == 1) g.W = plain_w(Wp, (dch[i + 1] + 15) & ~15, r * cpad);
            g.Y = dbuf[i + 1];
            if (i == 2) g.Pout = b.Zz;                                 // d2 as planes too: the input of downs[3]
            g.b_plane = (int64_t)(alive_planes_bytes((int64_t)N * len, dch[i], 2) / 4);      // elements per plane
            g.b_win = (int64_t)len * 32;
            g.b_row = r * 32;
            g.b_cblk = cpad / 32;
            g.b_blk = g.b_plane / cpad * 32;                                              // padded rows of the buffer x 32
            RUN(alive_gemm_planes(&g, stream));
        } else {
            AliveConv d = conv_desc(W, bb, dbuf[i], N, dch[i], len, dch[i + 1], r, r, 1, 0, 0, len / r, dbuf[i + 1]);
            if (i == 1 && b.Pa != nullptr) {                     // d1 also as the planes downs[2] reads (batch path)
                d.Yp = b.U;
                d.yp_planes = (decoder_bf16_mask() & 4) ? 1 : 2;
            }
            RUN(alive_conv1d(&d, stream));
        }
        len /= r;
    }
    const float* mW = t.next(); const float* mb = t.next();
    {   // mid CausalConv1d(256,256,5) + skips[3]   (decoder.py:190-191)
        AliveConv d = split(conv_desc(mW, mb, b.d3, N, 256, Lf, 256, 5, 1, 1, 4, 1, Lf, b.m));
        if ((decoder_bf16_mask() & 4) && (int64_t)N * Lf > 96) d = plain(d);       // (c)
        d.skip = b.d3;
        RUN(alive_conv1d(&d, stream));
    }
    const float* upW[4]; const float* upb[4];
    const float* upWc = nullptr; const float* upbc = nullptr;
    for (int i = 0; i < 4; ++i) {
        upW[i] = t.next(); upb[i] = t.next();
        if (F_MODE[i] == 1) { upWc = t.next(); upbc = t.next(); }
    }
    const float* skips[4] = {b.d2, b.d1, b.d0, nullptr};
    const float* cur = b.m;
    int cin = 256, L = Lf, film_off = 0;
    for (int s = 0; s < 4; ++s) {
        const int C = F_CH[s], r = F_UP[s];
        // (a) in decoder precision mode 1, batch path: the 256-channel block in one kernel (filter_big.hip), U -> Hh like the finer scales.
        // (The rule reads the signal's length only: a window's samples do not depend on how many windows share the call.)
        const bool fused256 = F_MODE[s] == 0 && (decoder_bf16_mask() & 1) != 0 && b.Pa != nullptr && fb256_enabled() && L * r >= 128;
        // (e) the same kernel for the 64-channel block (tiles of 512 columns); its input conv goes into the transposed conv too
        const bool fused64s = F_MODE[s] == 1 && (decoder_bf16_mask() & 16) != 0 && b.Pa != nullptr && fb64s_enabled() && L * r >= 512;
        {   // ConvTranspose1d(cin, C, r, r): rows = (co, j).  Unfused scale: the weights are ups[s] x input_conv (module/_pack.py), the
            // output is the block's residual stream itself
            AliveConv d = conv_desc(fused64s ? upWc : upW[s], fused64s ? upbc : upb[s], cur, N, cin, L, C * r, 1, 1, 1, 0, 0, L,
                                    F_MODE[s] == 0 && !fused256 ? b.Hh : b.U);
            d.up = r;
            if (F_SPLIT[s]) d = split(d);
            // (experiment, mask bit 8, off: the two transposed convs are bound by their stores -- 170.1 +- 0.3 ms per step either way, and the
            // fixture error goes 1.40e-5 -> 1.79e-5)
            if (F_SPLIT[s] && (decoder_bf16_mask() & 8) && (int64_t)N * L > 96) d = plain(d);
            RUN(alive_conv1d(&d, stream));
        }
        L *= r;
        if (F_MODE[s] != 0) {   // whole FilterBlock (+ skip) in one kernel: U -> Hh
            if (F_MODE[s] == 2) {
                const float* wpack = t.next();
                RUN(alive_filter_block_small_range(b.U, N, C, L, wpack, b.film, FILM_ROWS, Lw_frames, film_off, f_begin * (L / Lf), f_begin, Lf,
                                                   skips[s], b.Hh, stream));
            } else {
                const float* w16 = t.next(); const float* bias = t.next();
                const void* w6[6]; const float* b6[6];
                for (int q = 0; q < 6; ++q) {
                    w6[q] = plain_w(t.next(), C, 5 * C);
                    b6[q] = t.next();
                }
                // (e) decoder precision mode 1: the block's six k5 convs on one fp16 plane
                if (fused64s)
                    RUN(alive_filter_block64s_fp16(b.U, N, L, w6, b6, b.film, FILM_ROWS, Lw_frames, film_off, f_begin * (L / Lf), f_begin, Lf, skips[s],
                                                   b.Hh, b.Zz, (int64_t)N * 64 * (Lw / 4) * 4, stream));
                else if ((decoder_bf16_mask() & 16) && b.Pa != nullptr)          // (batch path only, like the other groups)
                    RUN(alive_filter_block64_range_fp16(b.U, N, L, w16, bias, b.film, FILM_ROWS, Lw_frames, film_off, f_begin * (L / Lf), f_begin, Lf,
                                                        skips[s], b.Hh, stream));
                else
                    RUN(alive_filter_block64_range(b.U, N, L, w16, bias, b.film, FILM_ROWS, Lw_frames, film_off, f_begin * (L / Lf), f_begin, Lf,
                                                   skips[s], b.Hh, stream));
            }
            film_off += 6 * 2 * C;
            cur = b.Hh;
            cin = C;
            continue;
        }
        if (fused256) {
            const void* w16[6]; const float* bias[6];
            for (int q = 0; q < 6; ++q) {
                w16[q] = plain_w(t.next(), C, 5 * C);
                bias[q] = t.next();
            }
            // (workspace: Zz, idle on this path -- N x 5120 Lf floats against N x (10 Lf / 128 + 1) x 48 KB)
            RUN(alive_filter_block256_fp16(b.U, N, L, w16, bias, b.film, FILM_ROWS, Lw_frames, film_off, f_begin * (L / Lf), f_begin, Lf, skips[s],
                                           b.Hh, b.Zz, (int64_t)N * 64 * (Lw / 4) * 4, stream));
            film_off += 6 * 2 * C;
            cur = b.Hh;
            cin = C;
            continue;
        }
        // The modulated tensors between the convs of this block (each written once, read once) travel plane-packed and
        // (k-blocked, rows = time) when the split kernel's batch form runs them (not the few-column streaming kernel): the consumer then
        // stages them by LDS-DMA instead of loading, splitting and storing fp32 (conv_split.hip, PLANES).  Same values bit for bit.
        const bool zplanes = F_SPLIT[s] && (int64_t)N * L > 96 && (L & 3) == 0 && C > 64 && C % 64 == 0 &&
                             (double)(ranged ? Lf : Lw_frames) / (double)L * 128.0 + 3.0 <= 20.0;
        auto set_z = [&](AliveConv& d, float* buf) { if (zplanes) d.Zp = buf; else d.Z = buf; };
        auto set_x = [&](AliveConv& d, const float* buf) { if (zplanes) { d.Xp = buf; d.X = nullptr; } };
        // FilterBlock.input_conv is part of the transposed conv above (Hh = its output); the input of blocks[0].c1 = gelu + FiLM of it
        const bool lowp = (decoder_bf16_mask() & 1) != 0 && (int64_t)N * L > 96;      // plain-fp16 convs: one plane travels between them
        RUN(alive_gelu_film_impl(b.Hh, N, C, L, b.film, FILM_ROWS, Lw_frames, film_off, film_off + C, ranged ? f_begin * (L / Lf) : 0,
                                 ranged ? f_begin : 0, ranged ? Lf : Lw_frames, zplanes ? nullptr : b.Zz, zplanes ? (void*)b.Zz : nullptr,
                                 lowp ? 1 : 2, stream));
        for (int j = 0; j < 3; ++j) {
            const int dil = 1 << j;
            const float* W1 = t.next(); const float* b1 = t.next(); const float* W2 = t.next(); const float* b2 = t.next();
            const int f1 = film_off + (j * 2) * 2 * C, f2 = film_off + (j * 2 + 1) * 2 * C;
            (void)f1;
            {   // c1: conv(Zz) -> only the modulated input of c2 is kept
                AliveConv d = conv_desc(W1, b1, b.Zz, N, C, L, C, 5, 1, dil, 4 * dil, 1, L, nullptr);
                set_x(d, b.Zz);
                set_z(d, b.Z2); d.film = b.film; d.film_rows = FILM_ROWS; d.Lf = Lw_frames;
            if (ranged) { d.film_t0 = f_begin * (L / Lf); d.film_f0 = f_begin; d.film_ld = Lf; }
                d.film_scale_row = f2; d.film_shift_row = f2 + C;
                if (F_SPLIT[s]) d = split(d);
                if (lowp) d = plain(d);
                RUN(alive_conv1d(&d, stream));
            }
            {   // c2: conv(Z2) + residual (+ U-Net skip after the last block) ; next block's c1 input
                AliveConv d = conv_desc(W2, b2, b.Z2, N, C, L, C, 5, 1, dil, 4 * dil, 1, L, b.Hh);
                set_x(d, b.Z2);
                d.residual = b.Hh;
                if (j == 2) d.skip = skips[s];
                if (j < 2) {
                    const int fn = film_off + ((j + 1) * 2) * 2 * C;
                    set_z(d, b.Zz); d.film = b.film; d.film_rows = FILM_ROWS; d.Lf = Lw_frames;
            if (ranged) { d.film_t0 = f_begin * (L / Lf); d.film_f0 = f_begin; d.film_ld = Lf; }
                    d.film_scale_row = fn; d.film_shift_row = fn + C;
                }
                if (F_SPLIT[s]) d = split(d);
                if (lowp) d = plain(d);
                RUN(alive_conv1d(&d, stream));
            }
        }
        film_off += 6 * 2 * C;
        cur = b.Hh;
        cin = C;
    }
    const float* oW = t.next(); const float* ob = t.next();
    return alive_filter_source_out(b.Hh, N, Lw, oW, ob, wave, stream);
}
}  // namespace

extern "C" int alive_decoder_precision(int mode) {
    if (mode == 1 || mode == 2) g_decoder_precision = mode;
    return decoder_precision();
}

int alive_f16_sat_conv_split(int), alive_f16_sat_gemm(int), alive_f16_sat_blocks(int), alive_f16_sat_conv(int), alive_f16_sat_filter_mid(int),
    alive_f16_sat_filter_big(int);
extern "C" int alive_f16_saturations(int reset) {
    const int a = alive_f16_sat_conv_split(reset), b = alive_f16_sat_gemm(reset), c = alive_f16_sat_blocks(reset), d = alive_f16_sat_conv(reset);
    const int e = alive_f16_sat_filter_mid(reset), f = alive_f16_sat_filter_big(reset);
    if (a < 0 || b < 0 || c < 0 || d < 0 || e < 0 || f < 0) return -1;
    const long long t = (long long)a + b + c + d + e + f;
    return (int)(t > 0x7fffffff ? 0x7fffffff : t);
}

int alive_f16_sat_conv_split_clear(void*), alive_f16_sat_gemm_clear(void*), alive_f16_sat_blocks_clear(void*), alive_f16_sat_conv_clear(void*),
    alive_f16_sat_filter_mid_clear(void*), alive_f16_sat_filter_big_clear(void*);
extern "C" int alive_f16_saturations_clear(void* stream) {
    const int r = alive_f16_sat_conv_split_clear(stream) | alive_f16_sat_gemm_clear(stream) | alive_f16_sat_blocks_clear(stream) |
                  alive_f16_sat_conv_clear(stream) | alive_f16_sat_filter_mid_clear(stream) | alive_f16_sat_filter_big_clear(stream);
    if (r != 0) {
        alive_set_error("alive_f16_saturations_clear: the runtime refused the asynchronous clear of a counter");
        return ALIVE_ERR_LAUNCH;
    }
    return ALIVE_OK;
}

extern "C" int alive_encoder_precision(int mode) {
    if (mode == 1 || mode == 2) g_encoder_precision = mode;
    return encoder_precision();
}

extern "C" int alive_decoder_forward(const float* const* w, const float* x_in, const float* f0, const float* phi_in, int crop0,
                                     int phi_col, int N, int Lf, float* wave, float* phi_out, void* ws, void* stream) {
    ALIVE_CHECK_ARG(w && x_in && f0 && wave && ws && N > 0, "alive_decoder_forward: bad args");
    ALIVE_CHECK_ARG(Lf >= 5, "alive_decoder_forward: needs at least 5 frames (reflect pad 4 on the bottleneck), got %d", Lf);
    DecBuffers b = dec_layout(ws, N, Lf);
    return decoder_run(w, x_in, f0, f0, phi_in, crop0, phi_col, N, Lf, Lf, 0, wave, phi_out, b, stream);
}

extern "C" int alive_decoder_forward_range(const float* const* w, const float* x_in, const float* f0, int N, int Lf, int f_begin,
                                           int n_frames, float* wave, void* ws, void* stream) {
    ALIVE_CHECK_ARG(w && x_in && f0 && wave && ws && N > 0, "alive_decoder_forward_range: bad args");
    ALIVE_CHECK_ARG(n_frames >= 5 && f_begin >= 0 && f_begin + n_frames <= Lf, "alive_decoder_forward_range: frames [%d, %d) of %d",
                    f_begin, f_begin + n_frames, Lf);
    // scratch: the range's own buffers, then an oscillator workspace for the whole window and the f0 slice
    DecBuffers b = dec_layout(ws, N, n_frames);
    Arena a((char*)ws + b.bytes);
    b.osc_ws = a.take<char>(alive_oscillator_workspace_bytes(N, NH, Lf));
    float* f0s = a.take<float>((size_t)N * n_frames);
    slice_frames_kernel<<<dim3(cdiv(n_frames, 256), N), 256, 0, (hipStream_t)stream>>>(f0, Lf, f_begin, n_frames, f0s);
    return decoder_run(w, x_in, f0s, f0, nullptr, 0, 0, N, n_frames, Lf, f_begin, wave, nullptr, b, stream);
}
