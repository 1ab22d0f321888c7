"""Operator-level Python wrappers over the C ABI (one per exported kernel family).

Used by the module classes and by the per-kernel parity tests.  All tensors are
fp32 contiguous HIP tensors; weights here are in the REFERENCE layout and are
packed on the fly (the network classes pack once at load time instead).
"""
import ctypes as C

import torch

from . import _native as nat
from ._pack import pack_conv, pack_conv_split, pack_conv_split_h, pack_convT, pack_convT_split

ACT = {None: 0, "gelu": 1, "exp": 2, "sin": 3}
_ws = nat.Workspace()


def _f(t):
    return None if t is None else t.contiguous().float()


def conv1d(x, weight, bias=None, stride=1, dilation=1, pad_left=0, pad_mode=0, out_len=None, act=None,
           post_add=None, ch_scale=None, residual=None, skip=None, film=None, film_scale_row=0, film_shift_row=0,
           want_raw=True, transposed=False, precision="fp32", x_planes=None, z_planes=False, y_planes=False):
    """Generic Conv1d / ConvTranspose1d(k == stride) through alive_conv1d.
    Returns (Y, Z): raw output (or None) and the gelu+FiLM modulated second output (or None).
    x_planes: the plane-packed form of x (to_planes(x, 2)) -- the split kernel then stages its input by LDS-DMA (AliveConv.Xp);
    z_planes: Z comes back plane-packed (AliveConv.Zp; a uint8 buffer like to_planes gives) instead of fp32;
    y_planes: the exact kernel also writes Y as planes (AliveConv.Yp) -- True / 2: two bf16 planes, 1: one fp16 plane; returned in Z's place."""
    x = _f(x)
    n, ci, tin = x.shape
    keep = []
    split = precision in ("bf16x3", "bf16x6", "fp16")          # "fp16": one MFMA per product on ONE fp16 plane (AliveConv.precision 3)
    planes = 3 if precision == "bf16x6" else 2
    if split and transposed:
        r = weight.shape[2]
        W, b = pack_convT_split(weight, bias)
        co_rows, kw, up = weight.shape[1] * r, 1, r
        stride_, tout = 1, tin
        co_out = weight.shape[1]
    elif split:
        W = pack_conv_split_h(weight)[2].contiguous() if precision == "fp16" else pack_conv_split(weight, planes)
        b = _f(bias)
        co_rows, kw, up = weight.shape[0], weight.shape[2], 1
        stride_ = stride
        tout = out_len if out_len is not None else (tin + pad_left - dilation * (kw - 1) - 1) // stride + 1
        co_out = co_rows
    elif transposed:
        r = weight.shape[2]
        W, b = pack_convT(weight, bias)
        co_rows, kw, up = weight.shape[1] * r, 1, r
        stride_, tout = 1, tin
        co_out = weight.shape[1]
    else:
        W = pack_conv(weight)
        b = _f(bias)
        co_rows, kw, up = weight.shape[0], weight.shape[2], 1
        stride_ = stride
        tout = out_len if out_len is not None else (tin + pad_left - dilation * (kw - 1) - 1) // stride + 1
        co_out = co_rows
    d = nat.AliveConv()
    d.W, d.bias, d.X = nat.ptr(W), nat.ptr(b), nat.ptr(x)
    d.N, d.Ci, d.Tin, d.Co, d.K_pad = n, ci, tin, co_rows, (W.shape[-1] if W.dim() == 2 else W.shape[1] * 32)
    d.precision, d.Ci_pad = (3 if precision == "fp16" else planes - 1, (ci + 31) // 32 * 32) if split else (0, 0)
    d.KW, d.stride, d.dil, d.pad_left, d.pad_mode = kw, stride_, dilation, pad_left, pad_mode
    d.Tout, d.up, d.act = tout, up, ACT[act]
    post_add, ch_scale, residual, skip, film = map(_f, (post_add, ch_scale, residual, skip, film))
    keep += [W, b, x, post_add, ch_scale, residual, skip, film]
    d.post_add, d.ch_scale, d.residual, d.skip = map(nat.ptr, (post_add, ch_scale, residual, skip))
    Y = torch.empty(n, co_out, tout * up, device=x.device) if want_raw else None
    Z = None
    if film is not None:
        if z_planes:
            Z = torch.empty(nat.lib().alive_planes_bytes(n * tout, co_out, 1 if precision == "fp16" else 2), dtype=torch.uint8, device=x.device)
        else:
            Z = torch.empty(n, co_out, tout, device=x.device)
        d.film, d.film_rows, d.Lf = nat.ptr(film), film.shape[1], film.shape[2]
        d.film_scale_row, d.film_shift_row = film_scale_row, film_shift_row
    d.Y = nat.ptr(Y)
    if z_planes:
        d.Zp = nat.ptr(Z)
    else:
        d.Z = nat.ptr(Z)
    if x_planes is not None:
        d.Xp, d.X = nat.ptr(x_planes), None
        keep.append(x_planes)
    if y_planes:
        ypl = 1 if y_planes == 1 and y_planes is not True else 2
        Z = torch.empty(nat.lib().alive_planes_bytes(n * tout, co_out, ypl), dtype=torch.uint8, device=x.device)
        d.Yp, d.yp_planes = nat.ptr(Z), ypl
    nat.check(nat.lib().alive_conv1d(C.byref(d), nat.stream()), "alive_conv1d")
    return Y, Z


def to_planes(x, planes=2):
    """fp32 [N][C][T] -> plane-packed buffer (uint8 tensor of alive_planes_bytes) for gemm_planes: 2 / 3 split-bf16 planes, or
    planes = 1: ONE fp16 plane (plain operands)."""
    x = _f(x)
    n, c, t = x.shape
    P = torch.empty(nat.lib().alive_planes_bytes(n * t, c, planes), dtype=torch.uint8, device=x.device)
    nat.check(nat.lib().alive_to_planes(nat.ptr(x), n, c, t, planes, nat.ptr(P), nat.stream()), "alive_to_planes")
    return P


def planes_to_float(P, n, c, t, planes):
    """inverse of to_planes (sum of the planes), for tests"""
    cols, cp = n * t, (c + 31) // 32 * 32
    cols_pad = (cols + 127) // 128 * 128
    v = P.view(torch.bfloat16).view(planes, cp // 32, cols_pad, 32).float().sum(0)        # k-blocked (csrc/planes_layout.h)
    v = v.permute(1, 0, 2).reshape(cols_pad, cp)[:cols, :c]
    return v.view(n, t, c).permute(0, 2, 1).contiguous()


def gelu_film(h, film, scale_row, shift_row, planes=False):
    """gelu(h) * interp(film[scale rows]) + interp(film[shift rows]) (decoder.py:112-117,130-132) through alive_gelu_film:
    fp32 [N][C][L], or the 2-plane k-blocked image (uint8 buffer like to_planes gives)."""
    h, film = _f(h), _f(film)
    n, c, l = h.shape
    z = None if planes else torch.empty_like(h)
    zp = torch.zeros(nat.lib().alive_planes_bytes(n * l, c, 2), dtype=torch.uint8, device=h.device) if planes else None
    nat.check(nat.lib().alive_gelu_film(nat.ptr(h), n, c, l, nat.ptr(film), film.shape[1], film.shape[2], scale_row, shift_row, 0, 0,
                                        film.shape[2], nat.ptr(z), nat.ptr(zp), nat.stream()), "alive_gelu_film")
    return zp if planes else z


def gemm_planes(P, n, t, weight, bias=None, planes=2, act=None, post_add=None, ch_scale=None, residual=None,
                want_fp32=True, want_planes=False):
    """1x1 conv on a plane-packed input through alive_gemm_planes.  Returns (Y fp32 [n][co][t] or None, Pout or None)."""
    co, ci = weight.shape[0], weight.shape[1]
    W = pack_conv_split_h(weight)[2].contiguous() if planes == 1 else pack_conv_split(weight, planes)     # planes = 1: one fp16 plane
    b, post_add, ch_scale, residual = map(_f, (bias, post_add, ch_scale, residual))
    d = nat.AliveGemm()
    d.W, d.bias, d.P = nat.ptr(W), nat.ptr(b), nat.ptr(P)
    d.N, d.T, d.Ci, d.Co, d.planes, d.act = n, t, ci, co, planes, ACT[act]
    d.post_add, d.ch_scale, d.residual = map(nat.ptr, (post_add, ch_scale, residual))
    Y = torch.empty(n, co, t, device=P.device) if want_fp32 else None
    Po = torch.empty(nat.lib().alive_planes_bytes(n * t, co, planes), dtype=torch.uint8, device=P.device) if want_planes else None
    d.Y, d.Pout = nat.ptr(Y), nat.ptr(Po)
    nat.check(nat.lib().alive_gemm_planes(C.byref(d), nat.stream()), "alive_gemm_planes")
    return Y, Po


def gemm_planes_argmax(P, n, t, weight, bias=None, planes=3):
    """argmax over the output channels of a 1x1 conv without storing its output (AliveGemm.act = 3 + alive_argmax_merge):
    float indices [n, 1, t] like F0Estimator.estimate."""
    co, ci = weight.shape[0], weight.shape[1]
    W = pack_conv_split(weight, planes)
    b = _f(bias)
    nblk = (co + 63) // 64
    val = torch.empty(nblk, n * t, device=P.device)
    idx = torch.empty(nblk, n * t, dtype=torch.int32, device=P.device)
    d = nat.AliveGemm()
    d.W, d.bias, d.P = nat.ptr(W), nat.ptr(b), nat.ptr(P)
    d.N, d.T, d.Ci, d.Co, d.planes, d.act = n, t, ci, co, planes, 3
    d.arg_val, d.arg_idx = nat.ptr(val), nat.ptr(idx)
    nat.check(nat.lib().alive_gemm_planes(C.byref(d), nat.stream()), "alive_gemm_planes")
    out = torch.empty(n, 1, t, device=P.device)
    nat.check(nat.lib().alive_argmax_merge(nat.ptr(val), nat.ptr(idx), nblk, n * t, nat.ptr(out), nat.stream()), "alive_argmax_merge")
    return out


def dwconv_norm(x, dw_w, dw_b, gain=None, offset=None, cond=None, scale_row=0, shift_row=0, eps=1e-4):
    x = _f(x)
    n, c, t = x.shape
    y = torch.empty_like(x)
    dw_w, dw_b, gain, offset, cond = map(_f, (dw_w.reshape(-1), dw_b, None if gain is None else gain.reshape(-1),
                                              None if offset is None else offset.reshape(-1), cond))
    rc = nat.lib().alive_dwconv_norm(nat.ptr(x), n, c, t, nat.ptr(dw_w), nat.ptr(dw_b), 0 if cond is None else 1,
                                     nat.ptr(gain), nat.ptr(offset), nat.ptr(cond),
                                     0 if cond is None else cond.shape[1], scale_row, shift_row, eps, nat.ptr(y),
                                     nat.stream())
    nat.check(rc, "alive_dwconv_norm")
    return y


def channel_norm(x, gain, offset, eps=1e-4):
    x = _f(x)
    n, c, t = x.shape
    y = torch.empty_like(x)
    g, o = _f(gain.reshape(-1)), _f(offset.reshape(-1))
    nat.check(nat.lib().alive_channel_norm(nat.ptr(x), n, c, t, nat.ptr(g), nat.ptr(o), eps, nat.ptr(y), nat.stream()),
              "alive_channel_norm")
    return y


def argmax_channels(x):
    x = _f(x)
    n, c, t = x.shape
    out = torch.empty(n, 1, t, device=x.device)
    nat.check(nat.lib().alive_argmax_channels(nat.ptr(x), n, c, t, nat.ptr(out), nat.stream()), "alive_argmax_channels")
    return out


def oscillator(amps, f0, phi=None, crop0=0, phi_col=None, seg=320, sample_rate=16000.0):
    """amps[N,H,Lf] (already exp'd), f0[N,1,Lf] -> wave[N,1,Lf*seg], phi_out[N,H] at phi_col (or None)."""
    amps, f0 = _f(amps), _f(f0)
    n, h, lf = amps.shape
    wave = torch.empty(n, 1, lf * seg, device=amps.device)
    phi_in = None if phi is None else _f(phi.reshape(n, h))
    phi_out = None if phi_col is None else torch.empty(n, h, device=amps.device)
    L = nat.lib()
    ws = _ws.get(L.alive_oscillator_workspace_bytes(n, h, lf), amps.device)
    rc = L.alive_oscillator(nat.ptr(amps), nat.ptr(f0), nat.ptr(phi_in), n, h, lf, seg, sample_rate, crop0,
                            0 if phi_col is None else phi_col, nat.ptr(wave), nat.ptr(phi_out), nat.ptr(ws), nat.stream())
    nat.check(rc, "alive_oscillator")
    return wave, phi_out


def pitch_transform_(f0, mode, f0_rate=1.0, pitch_shift=0.0, intonation=1.0):
    """in place on f0[N,1,T]; mode 0 = inference.py:119-130, mode 1 = realtime_inference.py:156-163"""
    n, _, t = f0.shape
    nat.check(nat.lib().alive_pitch_transform(nat.ptr(f0), n, t, mode, f0_rate, pitch_shift, intonation, nat.stream()),
              "alive_pitch_transform")
    return f0


def filter_block_small(x, sd, prefix, film, film_off, skip=None):
    """fused FilterBlock for C in (8, 16): x[N,C,L], reference-layout weights sd[prefix + ...], film[N,rows,Lf]."""
    from ._pack import pack_filter_small
    x, film, skip = _f(x), _f(film), _f(skip)
    n, c, l = x.shape
    w = pack_filter_small(sd, prefix).to(x.device)
    assert w.numel() == nat.lib().alive_filter_block_small_weights(c)
    out = torch.empty_like(x)
    nat.check(nat.lib().alive_filter_block_small(nat.ptr(x), n, c, l, nat.ptr(w), nat.ptr(film), film.shape[1], film.shape[2],
                                                 film_off, nat.ptr(skip), nat.ptr(out), nat.stream()), "alive_filter_block_small")
    return out


def filter_block64(x, sd, prefix, film, film_off, skip=None, plain=False):
    """fused FilterBlock for C = 64 (split-bf16 MFMA): x[N,64,L], reference-layout weights sd[prefix + ...].
    plain: the six k5 convs on one fp16 plane per operand (alive_filter_block64_range_fp16; decoder precision mode 1)."""
    from ._pack import pack_filter_mid
    x, film, skip = _f(x), _f(film), _f(skip)
    n, c, l = x.shape
    w, b = pack_filter_mid(sd, prefix)
    w, b = w.to(x.device), b.to(x.device)
    assert w.numel() == nat.lib().alive_filter_block64_weights()
    out = torch.empty_like(x)
    if plain:
        nat.check(nat.lib().alive_filter_block64_range_fp16(nat.ptr(x), n, l, nat.ptr(w), nat.ptr(b), nat.ptr(film), film.shape[1], film.shape[2],
                                                            film_off, 0, 0, film.shape[2], nat.ptr(skip), nat.ptr(out), nat.stream()),
                  "alive_filter_block64_range_fp16")
        return out
    nat.check(nat.lib().alive_filter_block64(nat.ptr(x), n, l, nat.ptr(w), nat.ptr(b), nat.ptr(film), film.shape[1], film.shape[2],
                                             film_off, nat.ptr(skip), nat.ptr(out), nat.stream()), "alive_filter_block64")
    return out


def filter_block256(x, sd, prefix, film, film_off, skip=None, t0=0, f0=0, frames=None):
    """fused FilterBlock for C = 256 (or 64: alive_filter_block64s_fp16) on plain fp16 operands WITHOUT its 1x1 input conv (the decoder
    composes that into the transposed conv that produces x; csrc/filter_big.hip, decoder precision mode 1): x[N,C,L] = the block's
    residual stream, reference-layout weights sd[prefix + '.blocks.j.c1 / c2 ...'].  t0 / f0 / frames: the window's place in a longer
    signal (alive_filter_block64_range)."""
    import ctypes
    from ._pack import pack_conv_split_h
    x, film, skip = _f(x), _f(film), _f(skip)
    n, c, l = x.shape
    assert c in (256, 64)
    entry, query = (("alive_filter_block256_fp16", "alive_filter_block256_workspace_bytes") if c == 256 else
                    ("alive_filter_block64s_fp16", "alive_filter_block64s_workspace_bytes"))
    keep, ws, bs = [], [], []
    for j in range(3):
        for cc in ("c1", "c2"):
            w = pack_conv_split_h(sd[f"{prefix}.blocks.{j}.{cc}.conv.conv.weight"].to(x.device))       # [3, K / 32, 256, 32]: slab 2 = fp16
            b = _f(sd[f"{prefix}.blocks.{j}.{cc}.conv.conv.bias"]).to(x.device)
            keep += [w, b]
            ws.append(nat.ptr(w[2]))
            bs.append(nat.ptr(b))
    nbytes = getattr(nat.lib(), query)(n, l)
    ws_buf = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    out = torch.empty_like(x)
    nat.check(getattr(nat.lib(), entry)(nat.ptr(x), n, l, (ctypes.c_void_p * 6)(*ws), (ctypes.c_void_p * 6)(*bs), nat.ptr(film),
                                                   film.shape[1], film.shape[2] if frames is None else frames, film_off, t0, f0,
                                                   film.shape[2], nat.ptr(skip), nat.ptr(out), nat.ptr(ws_buf), nbytes, nat.stream()),
              entry)
    return out


def filter_source_in(src, w_in, b_in, w_d, b_d):
    """downs[0](source_in(src)) in one streaming kernel: src[N,1,Lw] -> [N,16,Lw/2] (reference-layout weights)"""
    src = _f(src)
    n, _, lw = src.shape
    args = [_f(t).reshape(-1) for t in (w_in, b_in, w_d, b_d)]
    out = torch.empty(n, 16, lw // 2, device=src.device)
    nat.check(nat.lib().alive_filter_source_in(nat.ptr(src), n, lw, *[nat.ptr(a) for a in args], nat.ptr(out), nat.stream()),
              "alive_filter_source_in")
    return out


def filter_source_out(h, w, b):
    """source_out: h[N,8,Lw] -> [N,1,Lw]"""
    h = _f(h)
    n, _, lw = h.shape
    w, b = _f(w).reshape(-1), _f(b).reshape(-1)
    out = torch.empty(n, 1, lw, device=h.device)
    nat.check(nat.lib().alive_filter_source_out(nat.ptr(h), n, lw, nat.ptr(w), nat.ptr(b), nat.ptr(out), nat.stream()),
              "alive_filter_source_out")
    return out


def dwconv_norm_planes(x, dw_w, dw_b, gain=None, offset=None, cond=None, scale_row=0, shift_row=0, eps=1e-4, planes=2):
    """dw conv k7 (or none: dw_w=None) + (Adaptive)ChannelNorm, one pass, plane-packed output for gemm_planes"""
    x = _f(x)
    n, c, t = x.shape
    dw_w, dw_b, gain, offset, cond = map(_f, (dw_w, dw_b, gain, offset, cond))
    P = torch.empty(nat.lib().alive_planes_bytes(n * t, c, planes), dtype=torch.uint8, device=x.device)
    nat.check(nat.lib().alive_dwconv_norm_planes(nat.ptr(x), n, c, t, nat.ptr(dw_w), nat.ptr(dw_b), 0 if cond is None else 1,
                                                 nat.ptr(gain), nat.ptr(offset), nat.ptr(cond), 0 if cond is None else cond.shape[1],
                                                 scale_row, shift_row, eps, planes, nat.ptr(P), nat.stream()), "alive_dwconv_norm_planes")
    return P


_front = {}          # (id(ce table), id(pe table)) -> (merged input weights, merged biases, the two tables)
_front_ws = nat.Workspace()


def front_end(wave, ce, pe, out=None):
    """wave [N, L] (16 kHz) -> (content features [N, 768, L // 320], f0 classes [N, 1, L // 320]) = ce(spectrogram(wave)) and
    pe.estimate(spectrogram(wave)), bitwise -- through ONE entry point that never writes an fp32 spectrogram (alive_front_end:
    the DFT GEMM's epilogue leaves the magnitudes as the plane-packed operand of the two input layers, which run as one GEMM).
    /root/reference/module/spectrogram.py:5-10, content_encoder.py:21-25, f0_estimator.py:22-34.
    out = (feat, f0) tensors to write into.  Shapes the fused path does not take (a handful of frames: the streaming ring;
    non-default network sizes) go through the three separate calls."""
    from .spectrogram import dft_basis, spectrogram
    wave = wave.contiguous().float()
    n, l = wave.shape
    t = l // 320
    if ce.generic or pe.generic or n * t < 96 or l % 8 != 0 or l <= 640:
        spec = spectrogram(wave)
        return ce(spec, out=None if out is None else out[0]), pe.estimate(spec, out=None if out is None else out[1])
    L = nat.lib()
    tc, tp = ce.table(), pe.table()
    key = (id(tc), id(tp))
    hit = _front.get(key)
    if hit is None or hit[2] is not tc or hit[3] is not tp:
        if len(_front) >= 4:
            _front.pop(next(iter(_front)))
        # plane-packed weights are [planes][K / 32][Co_pad][32]: concatenating along the rows is concatenating dim 2
        w = torch.cat([tc.tensor("input.W"), tp.tensor("input.W")], dim=2).contiguous()
        b = torch.cat([tc.tensor("input.b"), tp.tensor("input.b")]).contiguous()
        torch.cuda.current_stream(wave.device).synchronize()      # (side streams read it: module/pipeline.py)
        hit = _front[key] = (w, b, tc, tp)
    feat = out[0] if out is not None else torch.empty(n, 768, t, device=wave.device)
    f0 = out[1] if out is not None else torch.empty(n, 1, t, device=wave.device)
    if tuple(feat.shape) != (n, 768, t) or tuple(f0.shape) != (n, 1, t) or not feat.is_contiguous() or not f0.is_contiguous() \
            or feat.dtype != torch.float32 or f0.dtype != torch.float32:
        raise ValueError("front_end: out must be contiguous fp32 ([N, 768, T], [N, 1, T]) tensors")
    ws = _front_ws.get(L.alive_front_end_workspace_bytes(n, l), wave.device)
    nat.check(L.alive_front_end(nat.ptr(dft_basis(wave.device)), tc.array, tp.array, nat.ptr(hit[0]), nat.ptr(hit[1]), nat.ptr(wave), n, l,
                                nat.ptr(feat), nat.ptr(f0), nat.ptr(ws), nat.stream()), "alive_front_end")
    return feat, f0


def decoder_precision(mode=0):
    """Arithmetic of the six k = 5 convs of the decoder's 256-channel FilterBlock on the batch path (alive_decoder_precision):
    1 = plain fp16 operands (default since round 5), 2 = two-plane split bf16 (rounds 1 - 4), 0 = query.  Returns the mode in force."""
    return int(nat.lib().alive_decoder_precision(int(mode)))


def encoder_precision(mode=0):
    """Arithmetic of the encoders' ConvNeXt pointwise convs on the batch path (alive_encoder_precision): 1 = fp16 split planes, three MFMAs
    per product (default since round 5), 2 = three bf16 planes, six MFMAs (rounds 1 - 4), 0 = query.  Returns the mode in force."""
    return int(nat.lib().alive_encoder_precision(int(mode)))


def f16_saturations(reset=False):
    """values saturated at +-65504 while an fp16 plane was written (alive_f16_saturations) on the current device since the last clear;
    SYNCHRONISES the device before it reads.  A read error raises (it must never pass for "no saturation")."""
    n = int(nat.lib().alive_f16_saturations(1 if reset else 0))
    if n < 0:
        raise RuntimeError("alive_f16_saturations: the device-side saturation counters could not be read (HIP runtime error)")
    return n


def f16_clear():
    """zeroes the saturation counters asynchronously, in the order of the current stream (alive_f16_saturations_clear)"""
    nat.check(nat.lib().alive_f16_saturations_clear(nat.stream()), "alive_f16_saturations_clear")


class Fp16Guard:
    """Range guard of the fp16 forms (alive_encoder_precision / alive_decoder_precision, modes 1) around one batch of work:

        out = Fp16Guard().run(lambda: work())

    clears the counters in stream order, runs `work`, reads the counters (one device synchronisation) and -- when a value left fp16's
    range -- REPEATS `work` with both precision modes at 2 (bf16 planes: fp32's range; the result is then the split-bf16 / bf16x6
    arithmetic of rounds 1-4, inside the same parity bars) instead of handing a saturated result on.  `fallbacks` counts the repeats;
    the modes in force before are restored.  Skipped altogether (no synchronisation) when both modes are already 2."""
    fallbacks = 0                       # process-wide tally, reported by bench.py / the CLIs

    def __init__(self, agree=None):
        # agree(n) -> n: under a sharded library every rank must take the same decision (the repeat runs collectives):
        # module/sharded.py passes the max over the ranks
        self.agree = agree

    def run(self, work):
        enc, dec = encoder_precision(0), decoder_precision(0)
        if enc == 2 and dec == 2:
            return work()
        f16_clear()
        out = work()
        n = f16_saturations(reset=True)
        if self.agree is not None:
            n = self.agree(n)
        if n == 0:
            return out
        import warnings
        warnings.warn(f"{n} activation value(s) left fp16's range in the fp16 forms of the encoder / decoder GEMMs: repeating this batch on "
                      "bf16 planes (ALIVE_ENCODER_PRECISION=2 ALIVE_DECODER_PRECISION=2 avoids the first attempt)", RuntimeWarning)
        Fp16Guard.fallbacks += 1
        try:
            encoder_precision(2)
            decoder_precision(2)
            out = work()
            if f16_saturations(reset=True) != 0:          # cannot happen: no fp16 plane is written in modes 2
                raise RuntimeError("fp16 saturations reported with both precision modes at 2")
        finally:
            encoder_precision(enc)
            decoder_precision(dec)
        return out
