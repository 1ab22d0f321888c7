"""Per-kernel parity: every exported operator of libalive_vc.so (through the C ABI) against the
CPU oracle / plain torch fp32 on the same seeded inputs.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import alive_oracle as O
from module import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def g(name, shape, seed=7, scale=1.0):
    return synthetic.gaussian(name, seed, shape, scale)


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30)).item()


@pytest.mark.parametrize("co,ci,kw,stride,dil,pad,mode,t", [
    (512, 641, 1, 1, 1, 0, 0, 450),      # CE input layer
    (1536, 512, 1, 1, 1, 0, 0, 37),      # pw1, ragged T
    (64, 512, 1, 1, 1, 0, 0, 24),        # to_amps
    (8, 1, 7, 1, 1, 3, 0, 2000),         # source_in (zero pad both sides)
    (1, 8, 7, 1, 1, 3, 0, 1999),         # source_out
    (16, 8, 2, 2, 1, 0, 0, 3200),        # down 0
    (256, 64, 8, 8, 1, 0, 0, 1600),      # down 2
    (256, 256, 10, 10, 1, 0, 0, 500),    # down 3
    (256, 256, 5, 1, 1, 4, 1, 50),       # mid causal conv (reflect)
    (64, 64, 5, 1, 2, 8, 1, 700),        # dilated causal
    (16, 16, 5, 1, 4, 16, 1, 1400),      # dilated causal, small C
    (8, 8, 5, 1, 4, 16, 1, 333),
    (24, 20, 3, 1, 1, 0, 0, 100),        # odd sizes
])
def test_conv1d_matches_torch(co, ci, kw, stride, dil, pad, mode, t):
    from module import ops
    x = g(f"cx{co}{ci}{kw}", (2, ci, t))
    w = g(f"cw{co}{ci}{kw}", (co, ci, kw), scale=1.0 / np.sqrt(ci * kw))
    b = g(f"cb{co}{ci}{kw}", (co,), scale=0.1)
    if mode == 1:
        ref = F.conv1d(F.pad(x, (pad, 0), mode="reflect"), w, b, stride=stride, dilation=dil)
    else:
        ref = F.conv1d(x, w, b, stride=stride, dilation=dil, padding=pad)
    y, _ = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), stride=stride, dilation=dil, pad_left=pad, pad_mode=mode,
                      out_len=ref.shape[2])
    assert y.shape == ref.shape
    assert relerr(y, ref) < 2e-6, relerr(y, ref)


def test_conv1d_k1_is_single_fma():
    """F0Encoder.c1 (decoder.py:16,21): K == 1 conv must be fma(w, x, b) bit for bit (it feeds sin)."""
    from module import ops
    f0 = (torch.arange(0, 4096, dtype=torch.float32) * 0.97).view(1, 1, -1)
    w = g("k1w", (512, 1, 1), scale=0.3)
    b = g("k1b", (512,), scale=0.5)
    y, _ = ops.conv1d(f0.to(DEV), w.to(DEV), b.to(DEV))
    fma = (w[:, 0].double() * f0[0].double() + b.double().view(-1, 1)).float().unsqueeze(0)   # fma(w, x, b), one rounding
    assert torch.equal(y.cpu(), fma)
    # ATen's CPU conv gives the same single-rounding result on the build host; other hosts may round the
    # product first, so against the host's F.conv1d only closeness is required
    torch.testing.assert_close(y.cpu(), F.conv1d(f0, w, b), rtol=3e-7, atol=1e-6)


@pytest.mark.parametrize("act", ["gelu", "exp", "sin"])
def test_conv1d_activations(act):
    from module import ops
    x = g("ax", (1, 40, 129))
    w = g("aw", (48, 40, 1), scale=0.15)
    b = g("ab", (48,), scale=0.1)
    ref = F.conv1d(x, w, b)
    ref = {"gelu": F.gelu, "exp": torch.exp, "sin": torch.sin}[act](ref)
    y, _ = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), act=act)
    torch.testing.assert_close(y.cpu(), ref, rtol=2e-5, atol=2e-6)


def test_conv1d_epilogue_scale_residual_skip():
    from module import ops
    x = g("ex", (2, 96, 77))
    w = g("ew", (32, 96, 1), scale=0.1)
    b = g("eb", (32,), scale=0.1)
    sc = g("es", (32,), scale=0.5)
    res = g("er", (2, 32, 77))
    sk = g("ek", (2, 32, 77))
    ref = F.conv1d(x, w, b) * sc.view(1, -1, 1) + res + sk
    y, _ = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), ch_scale=sc.to(DEV), residual=res.to(DEV), skip=sk.to(DEV))
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("ci,co,r,t", [(256, 256, 10, 45), (256, 64, 8, 450), (64, 16, 2, 999), (16, 8, 2, 2000), (32, 12, 4, 333), (32, 4, 4, 333)])
def test_conv_transpose(ci, co, r, t):
    from module import ops
    x = g(f"tx{ci}{r}", (2, ci, t))
    w = g(f"tw{ci}{r}", (ci, co, r), scale=1.0 / np.sqrt(ci))
    b = g(f"tb{ci}{r}", (co,), scale=0.1)
    ref = F.conv_transpose1d(x, w, b, stride=r)
    y, _ = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), transposed=True)
    assert y.shape == ref.shape
    assert relerr(y, ref) < 2e-6


def test_modulated_chain_matches_oracle(golden_dir):
    """input_conv -> (gelu, FiLM, causal conv) x2 + residual, i.e. FilterBlock with one res block,
    through the dual-output epilogue (decoder.py:112-134,146-150)."""
    from module import ops
    z = np.load(os.path.join(golden_dir, "blk_filter_block.npz"))
    sd = {"n." + k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    x, c = torch.from_numpy(z["x"]), torch.from_numpy(z["c"])
    ref = torch.from_numpy(z["y"])
    C = x.shape[1]
    ws, bs, post = [], [], []
    for j in range(3):
        for cc in ("c1", "c2"):
            p = f"n.blocks.{j}.{cc}"
            ws += [sd[p + ".to_scale.weight"], sd[p + ".to_shift.weight"]]
            bs += [sd[p + ".to_scale.bias"], sd[p + ".to_shift.bias"]]
            post += [torch.ones(C), torch.zeros(C)]
    film, _ = ops.conv1d(c.to(DEV), torch.cat(ws, 0).to(DEV), torch.cat(bs, 0).to(DEV), post_add=torch.cat(post).to(DEV))
    h, zz = ops.conv1d(x.to(DEV), sd["n.input_conv.weight"].to(DEV), sd["n.input_conv.bias"].to(DEV), film=film,
                       film_scale_row=0, film_shift_row=C)
    for j in range(3):
        d = 2 ** j
        p = f"n.blocks.{j}"
        _, z2 = ops.conv1d(zz, sd[p + ".c1.conv.conv.weight"].to(DEV), sd[p + ".c1.conv.conv.bias"].to(DEV), dilation=d,
                           pad_left=4 * d, pad_mode=1, out_len=x.shape[2], film=film, want_raw=False,
                           film_scale_row=(2 * j + 1) * 2 * C, film_shift_row=(2 * j + 1) * 2 * C + C)
        nxt = (2 * j + 2) * 2 * C
        h, zz = ops.conv1d(z2, sd[p + ".c2.conv.conv.weight"].to(DEV), sd[p + ".c2.conv.conv.bias"].to(DEV), dilation=d,
                           pad_left=4 * d, pad_mode=1, out_len=x.shape[2], residual=h,
                           film=film if j < 2 else None, film_scale_row=nxt, film_shift_row=nxt + C)
    assert relerr(h, ref) < 5e-6, relerr(h, ref)
    assert relerr(h, O.filter_block(dict(sd), "n", x, c)) < 5e-6


@pytest.mark.parametrize("c,t,adaptive", [(512, 450, False), (256, 37, False), (512, 24, True), (32, 19, True)])
def test_dwconv_norm(c, t, adaptive):
    from module import ops
    x = g(f"nx{c}{t}", (2, c, t))
    sd = {"n.dw_conv.weight": g("ndw", (c, 1, 7), scale=0.3), "n.dw_conv.bias": g("ndb", (c,), scale=0.1)}
    y0 = F.conv1d(x, sd["n.dw_conv.weight"], sd["n.dw_conv.bias"], padding=3, groups=c)
    if adaptive:
        cond = g("ncond", (2, 2 * c + 5, t))
        ref = O.channel_stats_normalise(y0) * cond[:, 3:3 + c] + cond[:, 3 + c:3 + 2 * c]
        y = ops.dwconv_norm(x.to(DEV), sd["n.dw_conv.weight"].to(DEV), sd["n.dw_conv.bias"].to(DEV), cond=cond.to(DEV),
                            scale_row=3, shift_row=3 + c)
    else:
        gain, off = 1 + g("ng", (1, c, 1), scale=0.1), g("no", (1, c, 1), scale=0.1)
        ref = O.channel_stats_normalise(y0) * gain + off
        y = ops.dwconv_norm(x.to(DEV), sd["n.dw_conv.weight"].to(DEV), sd["n.dw_conv.bias"].to(DEV), gain=gain.to(DEV),
                            offset=off.to(DEV))
    torch.testing.assert_close(y.cpu(), ref, rtol=2e-5, atol=2e-5)


def test_channel_norm_and_argmax():
    from module import ops
    x = g("cnx", (3, 256, 101))
    gain, off = 1 + g("cng", (1, 256, 1), scale=0.1), g("cno", (1, 256, 1), scale=0.1)
    ref = O.channel_stats_normalise(x) * gain + off
    torch.testing.assert_close(ops.channel_norm(x.to(DEV), gain.to(DEV), off.to(DEV)).cpu(), ref, rtol=2e-5, atol=2e-5)
    lg = g("amx", (2, 4096, 77))
    lg[0, 100, 5] = lg[0, 900, 5] = 50.0          # tie -> first index
    am = ops.argmax_channels(lg.to(DEV)).cpu()
    assert torch.equal(am, torch.argmax(lg, dim=1).float().unsqueeze(1))


@pytest.mark.parametrize("name", ["blk_oscillator", "blk_oscillator_carry"])
def test_oscillator_golden(golden_dir, name):
    from module import ops
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    sd = {"n." + k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    x, f0 = torch.from_numpy(z["x"]), torch.from_numpy(z["f0"])
    amps = torch.exp(F.conv1d(x, sd["n.to_amps.weight"], sd["n.to_amps.bias"]))
    carry = name.endswith("carry")
    phi = torch.from_numpy(z["phi_in"]) if carry else None
    crop0 = int(z["crop0"]) if carry else 0
    col = 4000 if carry else 3000
    wave, phi_out = ops.oscillator(amps.to(DEV), f0.to(DEV), phi=None if phi is None else phi.to(DEV), crop0=crop0,
                                   phi_col=col)
    ow, oph, dbg = O.harmonic_oscillator(sd, "n", x, f0, phi=0 if phi is None else phi, crop0=crop0, return_debug=True)
    err = (wave.cpu() - ow).abs().max().item()
    assert err < 2e-6, err
    torch.testing.assert_close(wave.cpu(), torch.from_numpy(z["wave"]), rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(phi_out.cpu(), oph[:, :, col], rtol=1e-4, atol=2e-5)


def test_oscillator_long_window_phase_exact():
    """450-frame window, 64 harmonics: the fp64-accumulate / fp32-round phase of every sample must
    reproduce the CPU cumsum; compared through the waveform at 1e-5."""
    from module import ops
    lf = 450
    f0 = (80.0 + 400.0 * torch.from_numpy(synthetic.uniform01("f0long", 3, lf)).float()).view(1, 1, lf)
    f0[0, 0, 100:120] = 0.0
    amps = torch.exp(g("ampl", (1, 64, lf), scale=0.5))
    wave, _ = ops.oscillator(amps.to(DEV), f0.to(DEV))
    formants = F.interpolate(f0 * (torch.arange(64) + 1).view(1, 64, 1), lf * 320, mode="linear")
    dt = torch.cumsum(formants / 16000, dim=2)
    dt = dt - dt[:, :, 0].unsqueeze(2)
    ref = (torch.sin(2 * np.pi * dt) * F.interpolate(amps, lf * 320, mode="linear")).mean(dim=1, keepdim=True)
    err = (wave.cpu() - ref).abs().max().item()
    assert err < 1e-5, err


@pytest.mark.parametrize("mode", [0, 1])
def test_pitch_transform(mode):
    from module import ops
    f0 = torch.from_numpy(np.round(synthetic.uniform01("pf0", 5, 2 * 450) * 600).astype(np.float32)).view(2, 1, 450)
    f0[0, 0, :40] = 0.0
    if mode == 0:
        ref = torch.cat([O.pitch_transform_offline(f0[i:i + 1].clone(), -3.0, 0.8, 0.5) for i in range(2)], 0)
        got = ops.pitch_transform_(f0.clone().to(DEV), 0, f0_rate=0.5, pitch_shift=-3.0, intonation=0.8)
    else:
        ref = O.pitch_transform_realtime(f0.clone() * 0.5, 2.0)
        got = ops.pitch_transform_(f0.clone().to(DEV), 1, f0_rate=0.5, pitch_shift=2.0)
    torch.testing.assert_close(got.cpu(), ref, rtol=3e-6, atol=1e-5)


@pytest.mark.parametrize("L", [1600, 2560, 144000])
def test_spectrogram(L):
    from module.spectrogram import spectrogram
    wav = torch.cat([synthetic.make_waveform(L, 9), synthetic.make_waveform(L, 10)], 0)
    ref = O.spectrogram(wav)
    got = spectrogram(wav.to(DEV)).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 2e-3 * ref.abs().max().item() * 1e-2 + 1e-3
    assert relerr(got, ref) < 1e-5


# ---- split-bf16 ("bf16x3") MFMA conv kernel: decoder GEMMs ----------------------------------------------
@pytest.mark.parametrize("co,ci,kw,dil,pad,mode,t", [
    (512, 768, 1, 1, 0, 0, 450),       # fe.input
    (1536, 512, 1, 1, 0, 0, 37),       # pw1, ragged T
    (64, 512, 1, 1, 0, 0, 24),         # to_amps (BM = 64 tile)
    (4128, 512, 1, 1, 0, 0, 130),      # FiLM projections, ragged Co
    (256, 256, 5, 1, 4, 1, 50),        # mid causal conv (reflect)
    (256, 256, 5, 4, 16, 1, 700),      # dilated causal, C = 256
    (64, 64, 5, 2, 8, 1, 1000),        # dilated causal, C = 64
    (40, 24, 3, 1, 2, 0, 100),         # odd sizes, zero pad
])
def test_conv1d_split_bf16(co, ci, kw, dil, pad, mode, t):
    from module import ops
    x = g(f"sx{co}{ci}{kw}", (2, ci, t))
    w = g(f"sw{co}{ci}{kw}", (co, ci, kw), scale=1.0 / np.sqrt(ci * kw))
    b = g(f"sb{co}{ci}{kw}", (co,), scale=0.1)
    if mode == 1:
        ref = F.conv1d(F.pad(x, (pad, 0), mode="reflect").double(), w.double(), b.double(), dilation=dil)
    else:
        ref = F.conv1d(F.pad(x, (pad, 0)).double(), w.double(), b.double(), dilation=dil)
    ref = ref[:, :, :t]
    y, _ = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), dilation=dil, pad_left=pad, pad_mode=mode, out_len=t,
                      precision="bf16x3")
    assert y.shape == ref.shape
    e = relerr(y, ref)
    assert e < 2e-5, e          # 2-term split: ~2^-16 per product; fp32 kernel is ~1e-7, plain bf16 would be ~3e-3


@pytest.mark.parametrize("ci,co,r,t", [(256, 256, 10, 45), (256, 64, 8, 450), (64, 40, 4, 1001), (96, 24, 16, 300)])
def test_conv_transpose_split_bf16(ci, co, r, t):
    from module import ops
    x = g(f"tsx{ci}{r}", (2, ci, t))
    w = g(f"tsw{ci}{r}", (ci, co, r), scale=1.0 / np.sqrt(ci))
    b = g(f"tsb{ci}{r}", (co,), scale=0.1)
    ref = F.conv_transpose1d(x.double(), w.double(), b.double(), stride=r)
    y, _ = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), transposed=True, precision="bf16x3")
    assert y.shape == ref.shape and relerr(y, ref) < 2e-5


def test_modulated_chain_split_bf16(golden_dir):
    """same FilterBlock chain as above on the split kernel (dual-output epilogue included)."""
    from module import ops
    z = np.load(os.path.join(golden_dir, "blk_filter_block.npz"))
    sd = {"n." + k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    x, c, ref = torch.from_numpy(z["x"]), torch.from_numpy(z["c"]), torch.from_numpy(z["y"])
    C = x.shape[1]
    ws, bs, post = [], [], []
    for j in range(3):
        for cc in ("c1", "c2"):
            p = f"n.blocks.{j}.{cc}"
            ws += [sd[p + ".to_scale.weight"], sd[p + ".to_shift.weight"]]
            bs += [sd[p + ".to_scale.bias"], sd[p + ".to_shift.bias"]]
            post += [torch.ones(C), torch.zeros(C)]
    kw = dict(precision="bf16x3")
    film, _ = ops.conv1d(c.to(DEV), torch.cat(ws, 0).to(DEV), torch.cat(bs, 0).to(DEV), post_add=torch.cat(post).to(DEV), **kw)
    h, zz = ops.conv1d(x.to(DEV), sd["n.input_conv.weight"].to(DEV), sd["n.input_conv.bias"].to(DEV), film=film,
                       film_scale_row=0, film_shift_row=C, **kw)
    for j in range(3):
        d = 2 ** j
        p = f"n.blocks.{j}"
        _, z2 = ops.conv1d(zz, sd[p + ".c1.conv.conv.weight"].to(DEV), sd[p + ".c1.conv.conv.bias"].to(DEV), dilation=d,
                           pad_left=4 * d, pad_mode=1, out_len=x.shape[2], film=film, want_raw=False,
                           film_scale_row=(2 * j + 1) * 2 * C, film_shift_row=(2 * j + 1) * 2 * C + C, **kw)
        nxt = (2 * j + 2) * 2 * C
        h, zz = ops.conv1d(z2, sd[p + ".c2.conv.conv.weight"].to(DEV), sd[p + ".c2.conv.conv.bias"].to(DEV), dilation=d,
                           pad_left=4 * d, pad_mode=1, out_len=x.shape[2], residual=h,
                           film=film if j < 2 else None, film_scale_row=nxt, film_shift_row=nxt + C, **kw)
    assert relerr(h, ref) < 1e-4, relerr(h, ref)


@pytest.mark.parametrize("l,lf,n,c", [(4500, 450, 2, 256), (370, 37, 3, 256), (1280, 128, 1, 256), (130, 13, 2, 256), (4490, 449, 9, 256),
                                      (36000, 450, 2, 64), (1200, 15, 3, 64), (408, 5, 2, 64), (5040, 63, 5, 64), (520, 6, 1, 64)])
def test_fused_filter_block_256(l, lf, n, c):
    """the 256-channel FilterBlock in one kernel (round 6, csrc/filter_big.hip: plain fp16 operands, a block sweeps a segment of a window in
    128-column tiles, contexts handed over through the workspace, warm-up tile per inner segment, reflection at a window's start) against
    the oracle's conv-by-conv evaluation (decoder.py:105-150) with an identity input conv: fp16 operand rounding (2^-12 per operand) is
    the whole difference.  Lengths that are no multiple of the tile or of 4, one tile, many blocks per window (n = 1) and few (n = 9).
    c = 64: the same kernel for decoder scale 1 (alive_filter_block64s_fp16: tiles of 512 columns, 80 samples per frame)."""
    from module import ops
    cond_ch = 24
    x = g(f"fb256x{l}", (n, c, l))
    cnd = g(f"fb256c{l}", (n, cond_ch, lf))
    skip = g(f"fb256s{l}", (n, c, l))
    sd = {"n.input_conv.weight": torch.eye(c).unsqueeze(-1).contiguous(), "n.input_conv.bias": torch.zeros(c)}
    ws, bs, post = [], [], []
    for j in range(3):
        for cc in ("c1", "c2"):
            p = f"n.blocks.{j}.{cc}"
            sd[p + ".conv.conv.weight"] = g(p + "w256", (c, c, 5), scale=0.5 / np.sqrt(c))
            sd[p + ".conv.conv.bias"] = g(p + "b256", (c,), scale=0.1)
            sd[p + ".to_scale.weight"] = g(p + "sw256", (c, cond_ch, 1), scale=0.1)
            sd[p + ".to_scale.bias"] = g(p + "sb256", (c,), scale=0.1)
            sd[p + ".to_shift.weight"] = g(p + "hw256", (c, cond_ch, 1), scale=0.1)
            sd[p + ".to_shift.bias"] = g(p + "hb256", (c,), scale=0.1)
            ws += [sd[p + ".to_scale.weight"], sd[p + ".to_shift.weight"]]
            bs += [sd[p + ".to_scale.bias"], sd[p + ".to_shift.bias"]]
            post += [torch.ones(c), torch.zeros(c)]
    ref = O.filter_block(sd, "n", x, cnd) + skip
    pad_rows = 5                                     # FiLM rows start at an offset inside a larger table, as in the decoder
    film, _ = ops.conv1d(cnd.to(DEV), torch.cat([torch.zeros(pad_rows, cond_ch, 1)] + ws, 0).to(DEV),
                         torch.cat([torch.zeros(pad_rows)] + bs, 0).to(DEV), post_add=torch.cat([torch.zeros(pad_rows)] + post).to(DEV))
    ops.f16_saturations(reset=True)
    out = ops.filter_block256(x.to(DEV), sd, "n", film, pad_rows, skip=skip.to(DEV))
    e = relerr(out, ref)
    assert e < 6e-4, e                               # (the 64-channel block's fp16 form: < 6e-4 on the same construction)
    assert ops.f16_saturations(reset=True) == 0
    for _ in range(2):                               # run to run, and window by window: a window's samples do not depend on how many
        assert torch.equal(ops.filter_block256(x.to(DEV), sd, "n", film, pad_rows, skip=skip.to(DEV)), out)       # windows share the call
    one = ops.filter_block256(x[:1].to(DEV), sd, "n", film[:1].contiguous(), pad_rows, skip=skip[:1].to(DEV))      # (other segmentation)
    assert torch.equal(one, out[:1])


@pytest.mark.parametrize("c", [256, 64])
def test_fused_filter_block_256_in_a_frame_range(c):
    """range mode (alive_filter_block64_range's t0 / f0 / film_ld): frames 40 .. 89 of a 128-frame signal with the FiLM rows of those frames
    only.  Past the reach of the window's start (its first 5 columns interpolate towards frame 39, which the table does not hold, and
    the reflected context reaches 4 (1 + 1 + 2 + 2 + 4 + 4) = 56 columns further) and short of its last frame (whose interpolation
    partner lies outside the table) the samples are those of the whole signal's run, bit for bit -- the interpolation
    coordinates are the signal's, and a column's sums do not depend on the tile it falls into."""
    from module import ops
    cond_ch, lf, f0, nf = 24, 128, 40, 50
    up = 10 if c == 256 else 80                      # samples per frame at the block's scale
    x = g(f"fb256rx{c}", (1, c, up * lf))
    cnd = g("fb256rc", (1, cond_ch, lf))
    sd = {}
    ws, bs, post = [], [], []
    for j in range(3):
        for cc in ("c1", "c2"):
            p = f"n.blocks.{j}.{cc}"
            sd[p + ".conv.conv.weight"] = g(p + "w256", (c, c, 5), scale=0.5 / np.sqrt(c))
            sd[p + ".conv.conv.bias"] = g(p + "b256", (c,), scale=0.1)
            ws += [g(p + "sw256", (c, cond_ch, 1), scale=0.1), g(p + "hw256", (c, cond_ch, 1), scale=0.1)]
            bs += [g(p + "sb256", (c,), scale=0.1), g(p + "hb256", (c,), scale=0.1)]
            post += [torch.ones(c), torch.zeros(c)]
    film, _ = ops.conv1d(cnd.to(DEV), torch.cat(ws, 0).to(DEV), torch.cat(bs, 0).to(DEV), post_add=torch.cat(post).to(DEV))
    whole = ops.filter_block256(x.to(DEV), sd, "n", film, 0)
    part = ops.filter_block256(x[:, :, up * f0:up * (f0 + nf)].contiguous().to(DEV), sd, "n", film[:, :, f0:f0 + nf].contiguous(), 0,
                               t0=up * f0, f0=f0, frames=lf)
    lo, hi = up // 2 + 1 + 56, up * nf - up - up // 2 - 1      # (first columns: towards frame f0 - 1; last: towards frame f0 + nf)
    assert torch.equal(part[:, :, lo:hi], whole[:, :, up * f0 + lo:up * f0 + hi])
    assert not torch.equal(part[:, :, :16], whole[:, :, up * f0:up * f0 + 16])             # (the reflected start is the window's own)


@pytest.mark.parametrize("c,l,lf", [(8, 4800, 15), (16, 2400, 15), (8, 144000, 450), (16, 1000, 5),
                                    (64, 1200, 15), (64, 36000, 450), (64, 400, 5), (64, 408, 5)])
def test_fused_filter_block_small(c, l, lf):
    """whole FilterBlock (input_conv + 3 res blocks, FiLM, reflect pads, halo recompute across tiles) in one kernel
    against the oracle's conv-by-conv evaluation (decoder.py:105-150).  C = 8, 16: exact fp32 (filter_small.hip);
    C = 64: split-bf16 (filter_mid.hip)."""
    from module import ops
    cond_ch = 24
    x = g(f"fbx{c}{l}", (2, c, l))
    cnd = g(f"fbc{c}{l}", (2, cond_ch, lf))
    skip = g(f"fbs{c}{l}", (2, c, l))
    sd = {"n.input_conv.weight": g("fbiw", (c, c, 1), scale=0.4), "n.input_conv.bias": g("fbib", (c,), scale=0.1)}
    ws, bs, post = [], [], []
    for j in range(3):
        for cc in ("c1", "c2"):
            p = f"n.blocks.{j}.{cc}"
            sd[p + ".conv.conv.weight"] = g(p + "w", (c, c, 5), scale=0.5 / np.sqrt(c))
            sd[p + ".conv.conv.bias"] = g(p + "b", (c,), scale=0.1)
            sd[p + ".to_scale.weight"] = g(p + "sw", (c, cond_ch, 1), scale=0.1)
            sd[p + ".to_scale.bias"] = g(p + "sb", (c,), scale=0.1)
            sd[p + ".to_shift.weight"] = g(p + "hw", (c, cond_ch, 1), scale=0.1)
            sd[p + ".to_shift.bias"] = g(p + "hb", (c,), scale=0.1)
            ws += [sd[p + ".to_scale.weight"], sd[p + ".to_shift.weight"]]
            bs += [sd[p + ".to_scale.bias"], sd[p + ".to_shift.bias"]]
            post += [torch.ones(c), torch.zeros(c)]
    ref = O.filter_block(sd, "n", x, cnd) + skip
    pad_rows = 5                                     # FiLM rows start at an offset inside a larger table, as in the decoder
    film, _ = ops.conv1d(cnd.to(DEV), torch.cat([torch.zeros(pad_rows, cond_ch, 1)] + ws, 0).to(DEV),
                         torch.cat([torch.zeros(pad_rows)] + bs, 0).to(DEV), post_add=torch.cat([torch.zeros(pad_rows)] + post).to(DEV))
    fused = ops.filter_block64 if c == 64 else ops.filter_block_small
    out = fused(x.to(DEV), {k: v.to(DEV) for k, v in sd.items()}, "n", film, pad_rows, skip=skip.to(DEV))
    e = relerr(out, ref)
    assert e < 4e-5, e             # split-bf16 products (~2^-16 each) at every scale since round 4 (C = 8 / 16 were exact f32 MFMA: 5e-6)
    for _ in range(3):                                # run-to-run determinism (see DESIGN.md 3.2b': the scheduling fence)
        again = fused(x.to(DEV), {k: v.to(DEV) for k, v in sd.items()}, "n", film, pad_rows, skip=skip.to(DEV))
        assert torch.equal(again, out)
    if c == 64:
        # round 5: the six k5 convs on ONE fp16 plane per operand (alive_filter_block64_range_fp16, decoder precision mode 1): 2^-12 per
        # operand instead of 2^-16 -- an order of magnitude from the split form, two below plain bf16 -- and as deterministic
        plain = ops.filter_block64(x.to(DEV), {k: v.to(DEV) for k, v in sd.items()}, "n", film, pad_rows, skip=skip.to(DEV), plain=True)
        ep = relerr(plain, ref)
        assert e < ep < 6e-4, (e, ep)
        for _ in range(20):
            again = ops.filter_block64(x.to(DEV), {k: v.to(DEV) for k, v in sd.items()}, "n", film, pad_rows, skip=skip.to(DEV), plain=True)
            assert torch.equal(again, plain)
        assert ops.f16_saturations() == 0


@pytest.mark.parametrize("co,ci,t", [(512, 641, 450), (1536, 512, 37), (4096, 256, 130), (768, 512, 450)])
def test_conv1d_split_bf16x6_is_fp32_grade(co, ci, t):
    """3-plane split (6 MFMAs per product): error at the level of fp32 rounding, so it may sit in front of the
    argmax of the f0 estimator and the top-k of the kNN like the exact fp32 kernel."""
    from module import ops
    x = g(f"s6x{co}{ci}", (2, ci, t))
    w = g(f"s6w{co}{ci}", (co, ci, 1), scale=1.0 / np.sqrt(ci))
    b = g(f"s6b{co}{ci}", (co,), scale=0.1)
    ref = F.conv1d(x.double(), w.double(), b.double())
    y6, _ = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), precision="bf16x6")
    y32, _ = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV))
    e6, e32 = relerr(y6, ref), relerr(y32, ref)
    print(f"bf16x6 {e6:.2e}  fp32-MFMA {e32:.2e}  torch-cpu-fp32 {relerr(F.conv1d(x, w, b), ref):.2e}")
    assert e6 < 4e-7 and e6 < 4 * e32 + 1e-8, (e6, e32)


@pytest.mark.parametrize("co,ci,n,t,prec", [(512, 641, 1, 8, "fp32"), (1536, 512, 1, 8, "bf16x6"), (4128, 512, 1, 24, "bf16x3"),
                                             (64, 512, 2, 5, "bf16x3"), (4096, 256, 1, 13, "bf16x6"), (40, 20, 3, 7, "fp32")])
def test_conv1d_skinny_streaming_shapes(co, ci, n, t, prec):
    """N*T <= 32 columns: the K-split skinny kernel (streaming path) for every weight format"""
    from module import ops
    x = g(f"kx{co}{ci}{t}", (n, ci, t))
    w = g(f"kw{co}{ci}{t}", (co, ci, 1), scale=1.0 / np.sqrt(ci))
    b = g(f"kb{co}{ci}{t}", (co,), scale=0.1)
    res = g(f"kr{co}{ci}{t}", (n, co, t))
    ref = F.gelu(F.conv1d(x.double(), w.double(), b.double())) + res.double()
    y, _ = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), act="gelu", residual=res.to(DEV), precision=prec)
    e = relerr(y, ref)
    assert e < (2e-5 if prec == "bf16x3" else 5e-7), e


def test_small_t_norm_and_argmax():
    """T <= 32: one block per frame (streaming variants of dwconv+norm, ChannelNorm, argmax)"""
    from module import ops
    for c, t in ((512, 8), (256, 5), (512, 24)):
        x = g(f"snx{c}{t}", (2, c, t))
        dw_w, dw_b = g("sndw", (c, 1, 7), scale=0.3), g("sndb", (c,), scale=0.1)
        gain, off = 1 + g("sng", (1, c, 1), scale=0.1), g("sno", (1, c, 1), scale=0.1)
        ref = O.channel_stats_normalise(F.conv1d(x, dw_w, dw_b, padding=3, groups=c)) * gain + off
        y = ops.dwconv_norm(x.to(DEV), dw_w.to(DEV), dw_b.to(DEV), gain=gain.to(DEV), offset=off.to(DEV))
        torch.testing.assert_close(y.cpu(), ref, rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(ops.channel_norm(x.to(DEV), gain.to(DEV), off.to(DEV)).cpu(),
                                   O.channel_stats_normalise(x) * gain + off, rtol=2e-5, atol=2e-5)
    lg = g("samx", (2, 4096, 8))
    lg[1, 77, 3] = lg[1, 3000, 3] = 60.0
    assert torch.equal(ops.argmax_channels(lg.to(DEV)).cpu(), torch.argmax(lg, dim=1).float().unsqueeze(1))


# ---- plane-packed GEMMs (frame-rate 1x1 convs of the ConvNeXt stacks) -----------------------------------
@pytest.mark.parametrize("planes", [2, 3])
@pytest.mark.parametrize("n,c,t", [(2, 641, 450), (1, 512, 37), (3, 40, 130)])
def test_to_planes_roundtrip(planes, n, c, t):
    from module import ops
    x = g(f"tp{n}{c}{t}", (n, c, t))
    P = ops.to_planes(x.to(DEV), planes)
    back = ops.planes_to_float(P, n, c, t, planes).cpu()
    # 2 planes carry 16 mantissa bits, 3 planes all 24
    assert (back - x).abs().max().item() <= (2.0 ** -15 if planes == 2 else 2.0 ** -22) * x.abs().max().item()
    cp, cols_pad = (c + 31) // 32 * 32, (n * t + 127) // 128 * 128
    raw = P.view(torch.bfloat16).view(planes, cp // 32, cols_pad, 32).float()              # k-blocked (csrc/planes_layout.h)
    raw = raw.permute(0, 2, 1, 3).reshape(planes, cols_pad, cp)
    assert raw[:, n * t:, :].abs().sum().item() == 0.0 and raw[:, :, c:].abs().sum().item() == 0.0      # zero padding


@pytest.mark.parametrize("planes,tol", [(2, 2e-5), (3, 4e-7)])
@pytest.mark.parametrize("co,ci,n,t", [
    (512, 641, 2, 450),       # encoder input convs, ragged K (641 -> 672)
    (1536, 512, 3, 37),       # pw1, ragged columns
    (512, 1536, 2, 450),      # pw2
    (4128, 512, 1, 130),      # FiLM projections, ragged Co
    (64, 512, 2, 24),         # to_amps: one partial row tile
    (4096, 256, 2, 200),      # f0 logits, K = 256 (8 steps)
    (40, 24, 3, 50),          # K = 32: a single step, fewer than the ring depth
    (256, 64, 1, 129),        # two steps
    (4096, 256, 10, 450),     # 1152 tiles: the persistent kernel (DMA ring running across the tile seams), ragged last column tile
    (1536, 512, 24, 450),     # 1020 / 1032 tiles around the persistence threshold
])
def test_gemm_planes_vs_float64(planes, tol, co, ci, n, t):
    from module import ops
    x = g(f"gpx{co}{ci}", (n, ci, t))
    w = g(f"gpw{co}{ci}", (co, ci, 1), scale=1.0 / np.sqrt(ci))
    b = g(f"gpb{co}{ci}", (co,), scale=0.1)
    res = g(f"gpr{co}{ci}", (n, co, t))
    sc = g(f"gps{co}{ci}", (co,), scale=0.5)
    ref = F.conv1d(x.double(), w.double(), b.double()) * sc.double().view(1, -1, 1) + res.double()
    P = ops.to_planes(x.to(DEV), planes)
    y, _ = ops.gemm_planes(P, n, t, w.to(DEV), b.to(DEV), planes=planes, ch_scale=sc.to(DEV), residual=res.to(DEV))
    assert y.shape == ref.shape
    e = relerr(y, ref)
    assert e < tol, e


@pytest.mark.parametrize("planes,tol", [(2, 3e-5), (3, 1e-6)])
def test_gemm_planes_chain_pw1_gelu_pw2(planes, tol):
    """pw1 -> GELU -> plane-packed hidden -> pw2 (+ layer scale + residual): the ConvNeXt MLP of common.py:57-62"""
    from module import ops
    n, c, h, t = 2, 512, 1536, 450
    x = g("gcx", (n, c, t))
    w1, b1 = g("gcw1", (h, c, 1), scale=1.0 / np.sqrt(c)), g("gcb1", (h,), scale=0.1)
    w2, b2 = g("gcw2", (c, h, 1), scale=1.0 / np.sqrt(h)), g("gcb2", (c,), scale=0.1)
    sc, res = g("gcs", (c,), scale=0.25), g("gcr", (n, c, t))
    hid = F.gelu(F.conv1d(x.double(), w1.double(), b1.double()))
    ref = F.conv1d(hid, w2.double(), b2.double()) * sc.double().view(1, -1, 1) + res.double()
    P = ops.to_planes(x.to(DEV), planes)
    hy, _ = ops.gemm_planes(P, n, t, w1.to(DEV), b1.to(DEV), planes=planes, act="gelu")
    _, Ph = ops.gemm_planes(P, n, t, w1.to(DEV), b1.to(DEV), planes=planes, act="gelu", want_fp32=False, want_planes=True)
    assert relerr(hy, hid) < tol
    assert relerr(ops.planes_to_float(Ph, n, h, t, planes), hid) < tol + (2e-5 if planes == 2 else 0.0)
    y, _ = ops.gemm_planes(Ph, n, t, w2.to(DEV), b2.to(DEV), planes=planes, ch_scale=sc.to(DEV), residual=res.to(DEV))
    e = relerr(y, ref)
    assert e < tol, e


@pytest.mark.parametrize("n,c,l,lf", [(2, 256, 4500, 450), (1, 256, 50, 5), (3, 64, 1201, 15), (1, 96, 640, 8)])
def test_gelu_film_equals_the_conv_second_output(n, c, l, lf):
    """alive_gelu_film (the input of a FilterBlock's first modulated conv since its input_conv is composed into the transposed conv
    in front, decoder.py:112-117,130-132,147,192-193) against float64, against the second output of a conv epilogue (bitwise: same
    arithmetic) and its plane image against alive_to_planes of the fp32 form"""
    from module import ops
    h = g(f"gfh{c}{l}", (n, c, l))
    film = g(f"gff{c}{l}", (n, 2 * c + 3, lf))
    # the interpolation coordinates are fp32 in the reference (ATen's upsample_linear1d on float tensors): interpolate in fp32, the rest in float64
    sc = F.interpolate(film[:, 3:3 + c], l, mode="linear").double()
    sh = F.interpolate(film[:, 3 + c:3 + 2 * c], l, mode="linear").double()
    ref = F.gelu(h.double()) * sc + sh
    z = ops.gelu_film(h.to(DEV), film.to(DEV), 3, 3 + c)
    assert relerr(z, ref) < 2e-6, relerr(z, ref)
    if c % 32 == 0:
        zp = ops.gelu_film(h.to(DEV), film.to(DEV), 3, 3 + c, planes=True)
        want = ops.to_planes(z, 2)
        cols, cols_pad = n * l, (n * l + 127) // 128 * 128
        a = zp.view(torch.bfloat16).view(2, c // 32, cols_pad, 32)[:, :, :cols].contiguous()
        e = want.view(torch.bfloat16).view(2, c // 32, cols_pad, 32)[:, :, :cols].contiguous()
        assert torch.equal(a.view(torch.int16), e.view(torch.int16))
    if l % 4 == 0:
        # a 1x1 conv with the identity as weight passes h through exactly on the fp32 kernel; its second output is the same formula
        _, z2 = ops.conv1d(h.to(DEV), torch.eye(c).unsqueeze(2).to(DEV), None, film=film.to(DEV), film_scale_row=3, film_shift_row=3 + c)
        assert torch.equal(z2, z)


def _digest_of(cmd, env):
    import subprocess, sys as _sys
    e = dict(os.environ); e.update(env)
    out = subprocess.run([_sys.executable] + cmd, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [ln for ln in out.stdout.splitlines() if ln.startswith("digest")][-1]


def test_gemm_results_do_not_depend_on_the_tile_order():
    """the row-group tile order of the plane GEMMs (gemm_planes.hip::tile_of, ALIVE_GEMM_RG) decides where and when a tile runs, never
    what it computes: every group size, the ragged last group (33 row tiles in groups of 5 / 7) and the old order give the same bits,
    on the one-tile kernels and on the persistent kernel with loader waves (the switches are read once per process: subprocesses)"""
    tool = os.path.join(ROOT, "tools", "run_gemm_once.py")
    for planes, variant in ((2, "0"), (3, "0"), (3, "1")):
        ds = {rg: _digest_of([tool, "512", "4128", str(planes), "1"], {"ALIVE_GEMM_RG": rg, "ALIVE_GEMM_VARIANT": variant})
              for rg in ("1", "5", "7", "64")}
        assert len(set(ds.values())) == 1, (planes, variant, ds)


@pytest.mark.parametrize("c,l,n,switch,forms", [(64, 800, 1, "ALIVE_FB64_NT", ("4", "2")), (64, 1208, 2, "ALIVE_FB64_NT", ("4", "2")),
                                                (16, 1600, 1, "ALIVE_FBS_PLANE", ("32768", "8192")),
                                                (8, 3200, 1, "ALIVE_FBS_PLANE", ("32768", "8192")),
                                                (8, 14400, 3, "ALIVE_FBS_PLANE", ("32768", "16384")),       # two blocks per CU (C = 8 default, round 5)
                                                (16, 7200, 3, "ALIVE_FBS_PLANE", ("32768", "16384"))])
def test_fused_filter_blocks_short_signal_tiles_are_bitwise_the_batch_tiles(c, l, n, switch, forms):
    """the short-signal tile forms of the fused FilterBlocks (filter_mid.hip: two column tiles per wave, with the drained first step of
    the reflecting block; filter_small.hip: 8-KB planes, and the 16-KB planes that put two blocks on a CU) compute what the batch tiles
    compute, bit for bit"""
    tool = os.path.join(ROOT, "tools", "run_fused_once.py")
    ds = [_digest_of([tool, str(c), str(l), str(n)], {switch: f}) for f in forms]
    assert ds[0] == ds[1], ds


def test_gemm_planes_argument_errors():
    from module import ops
    P = ops.to_planes(g("gex", (1, 32, 8)).to(DEV), 2)
    with pytest.raises(ValueError):
        ops.gemm_planes(P, 1, 8, torch.zeros(8, 32, 1, device=DEV), planes=4)
    with pytest.raises(ValueError):
        ops.gemm_planes(P, 1, 8, torch.zeros(8, 32, 1, device=DEV), want_fp32=False, want_planes=False)
    with pytest.raises(ValueError):
        ops.gemm_planes_argmax(P, 1, 8, torch.zeros(8, 32, 1, device=DEV), planes=2)      # argmax mode: 3 planes only


def test_gemm_planes_both_outputs_agree():
    """fp32 and plane-packed output of one launch (a decoder down conv feeds a skip and the next GEMM)"""
    from module import ops
    n, ci, co, t = 2, 96, 200, 333
    x, w, b = g("gbx", (n, ci, t)), g("gbw", (co, ci, 1), scale=0.1), g("gbb", (co,), scale=0.1)
    P = ops.to_planes(x.to(DEV), 2)
    y, Po = ops.gemm_planes(P, n, t, w.to(DEV), b.to(DEV), want_fp32=True, want_planes=True)
    y1, _ = ops.gemm_planes(P, n, t, w.to(DEV), b.to(DEV))
    assert torch.equal(y, y1)
    back = ops.planes_to_float(Po, n, co, t, 2)
    assert (back - y).abs().max().item() <= 2.0 ** -15 * y.abs().max().item()


@pytest.mark.parametrize("n,ci,co,t", [(2, 256, 4096, 450), (1, 64, 100, 130), (3, 32, 64, 7), (128, 256, 4096, 45)])
def test_gemm_planes_argmax_equals_argmax_of_the_stored_product(n, ci, co, t):
    """act = 3 keeps one candidate per 64-row block and column; merged, it is the argmax of the very values the Y path stores
    (same accumulators, same bias add) -- first index on ties, planted below"""
    from module import ops
    x = g(f"gax{co}", (n, ci, t))
    w = g(f"gaw{co}", (co, ci, 1), scale=1.0 / np.sqrt(ci))
    w[co // 2 + 1] = w[3]                      # duplicate rows: exact ties, the lower row must win
    w[co - 1] = w[3]
    b = g(f"gab{co}", (co,), scale=0.01)
    b[co // 2 + 1] = b[3]
    b[co - 1] = b[3]
    P = ops.to_planes(x.to(DEV), 3)
    y, _ = ops.gemm_planes(P, n, t, w.to(DEV), b.to(DEV), planes=3)
    want = torch.argmax(y, dim=1, keepdim=True).float()
    first = (y == y.max(dim=1, keepdim=True).values).float().argmax(dim=1, keepdim=True).float()     # first index of the max
    got = ops.gemm_planes_argmax(P, n, t, w.to(DEV), b.to(DEV))
    assert torch.equal(got, first)
    assert (got != want).float().mean().item() < 0.01          # torch.argmax may pick another of the tied rows


# ---- audio edges: resampler, gain, int16 conversions on the device -----------------------------------
@pytest.mark.parametrize("orig,new,L", [(24000, 16000, 24000), (16000, 24000, 16001), (48000, 16000, 4803), (44100, 16000, 22050),
                                        (16000, 22050, 3000)])
def test_resample_matches_torch_formulation(orig, new, L):
    from module import audio_io
    x = synthetic.make_waveform(L, 3)
    x = torch.cat([x, 0.5 * synthetic.make_waveform(L, 4)], 0)          # two channels
    ref = O.resample(x, orig, new)
    got = audio_io.resample(x.to(DEV), orig, new).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())
    g2 = audio_io.resample(x.to(DEV), orig, new, pre_gain_db=-3.0, post_gain_db=1.5).cpu()
    ref2 = O.gain(O.resample(O.gain(x, -3.0), orig, new), 1.5)
    assert (g2 - ref2).abs().max().item() < 2e-6 * max(1.0, ref2.abs().max().item())


def test_pcm16_conversions_are_the_reference_casts():
    from module import audio_io
    pcm = torch.arange(-32768, 32768, 7, dtype=torch.int16)
    f = audio_io.pcm16_to_float(pcm.to(DEV)).cpu()
    assert torch.equal(f, pcm.float() / 32768)
    w = torch.tensor([0.0, 0.5, -0.5, 0.99999, -1.0, 1.0, 1.2, -1.3, 3.0e-5, -3.0e-5, 0.123456], dtype=torch.float32)
    got = audio_io.float_to_pcm16(w.to(DEV)).cpu().numpy()
    with np.errstate(invalid="ignore"):
        want = (w.numpy() * 32768).astype(np.int32).astype(np.int16)     # truncate toward zero, low 16 bits: numpy's astype on x86
    assert np.array_equal(got, want), (got, want)


@pytest.mark.parametrize("n,lw", [(2, 9600), (1, 1600), (3, 322)])
def test_filter_edge_kernels(n, lw):
    """source_in + downs[0] fused, and source_out (decoder.py:164,182,186-188,194), against torch conv1d"""
    from module import ops
    src = g(f"fes{lw}", (n, 1, lw))
    w_in, b_in = g("fewi", (8, 1, 7), scale=0.3), g("febi", (8,), scale=0.1)
    w_d, b_d = g("fewd", (16, 8, 2), scale=0.25), g("febd", (16,), scale=0.1)
    ref = F.conv1d(F.conv1d(src.double(), w_in.double(), b_in.double(), padding=3), w_d.double(), b_d.double(), stride=2)
    got = ops.filter_source_in(src.to(DEV), w_in.to(DEV), b_in.to(DEV), w_d.to(DEV), b_d.to(DEV))
    assert got.shape == ref.shape and relerr(got, ref) < 5e-7
    h = g(f"feh{lw}", (n, 8, lw))
    w_o, b_o = g("fewo", (1, 8, 7), scale=0.2), g("febo", (1,), scale=0.1)
    ref = F.conv1d(h.double(), w_o.double(), b_o.double(), padding=3)
    got = ops.filter_source_out(h.to(DEV), w_o.to(DEV), b_o.to(DEV))
    assert got.shape == ref.shape and relerr(got, ref) < 5e-7


def test_div16000_is_the_ieee_quotient():
    """the oscillator divides by the sample rate with one multiply and two fmas (csrc/oscillator.hip::div_rate); the
    reference's `formants / 16000` is an IEEE division.  Exhaustive over every fp32 in [2^-24, 2^24)."""
    import ctypes as C
    from module import _native as nat
    fn = nat.lib().alive_debug_div16000_mismatches
    fn.restype, fn.argtypes = C.c_int, [C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
    bad = torch.zeros(1, dtype=torch.int32, device=DEV)
    first, last = 0x33800000, 0x4B800000          # 2^-24 .. 2^24
    nat.check(fn(first, last - first, bad.data_ptr(), nat.stream()))
    assert bad.item() == 0


@pytest.mark.parametrize("co,ci,kw,stride,dil,pad,mode,tin,tout,prec", [
    (1282, 1, 1280, 320, 1, 640, 2, 2560, 8, "fp32"),        # the DFT of an 8-frame ring (reflect both)
    (256, 64, 8, 8, 1, 0, 0, 640, 80, "fp32"),               # downs[2]
    (256, 256, 10, 10, 1, 0, 0, 80, 8, "fp32"),              # downs[3]
    (256, 256, 5, 1, 4, 16, 1, 80, 80, "bf16x3"),            # dilated causal k5, reflect left
    (256, 256, 5, 1, 1, 4, 1, 8, 8, "bf16x3"),               # mid conv on 8 frames
    (24, 40, 3, 1, 2, 4, 0, 31, 31, "fp32"),                 # odd sizes, zero pad
])
def test_conv1d_few_columns_any_geometry(co, ci, kw, stride, dil, pad, mode, tin, tout, prec):
    """<= 96 GEMM columns (a streaming step): the K-split skinny kernel serves every conv geometry of alive_conv1d"""
    from module import ops
    x = g(f"fcx{co}{ci}{kw}", (1, ci, tin))
    w = g(f"fcw{co}{ci}{kw}", (co, ci, kw), scale=1.0 / np.sqrt(ci * kw))
    b = g(f"fcb{co}{ci}{kw}", (co,), scale=0.1)
    if mode == 2:
        xp = F.pad(x, (pad, pad), mode="reflect")
    elif mode == 1:
        xp = F.pad(x, (pad, 0), mode="reflect")
    else:
        xp = F.pad(x, (pad, 0))
    ref = F.conv1d(xp.double(), w.double(), b.double(), stride=stride, dilation=dil)[:, :, :tout]
    y, _ = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), stride=stride, dilation=dil, pad_left=pad, pad_mode=mode, out_len=tout,
                      precision=prec)
    assert y.shape == ref.shape
    e = relerr(y, ref)
    assert e < (2e-5 if prec == "bf16x3" else 5e-7), e


@pytest.mark.parametrize("planes", [2, 3])
@pytest.mark.parametrize("n,c,t,adaptive,dw", [(2, 512, 450, False, True), (3, 256, 130, False, True), (2, 512, 70, True, True),
                                              (2, 256, 450, False, False)])
def test_dwconv_norm_planes_single_pass(planes, n, c, t, adaptive, dw):
    """dw conv + (Adaptive)ChannelNorm straight to planes == the fp32 kernel followed by alive_to_planes (common.py:20-41,55-56)"""
    from module import ops
    x = g(f"dnp{c}{t}", (n, c, t))
    w, b = (g("dnpw", (c, 1, 7), scale=0.3), g("dnpb", (c,), scale=0.1)) if dw else (None, None)
    gain, off = g("dnpg", (c,), scale=0.5) + 1.0, g("dnpo", (c,), scale=0.2)
    cond = g("dnpc", (n, 2 * c + 5, t)) if adaptive else None
    y = F.conv1d(x.double(), w.double(), b.double(), padding=3, groups=c) if dw else x.double()
    mu, sd = y.mean(dim=1, keepdim=True), y.std(dim=1, keepdim=True) + 1e-4
    ref = (y - mu) / sd
    ref = ref * cond[:, 5:5 + c].double() + cond[:, 5 + c:5 + 2 * c].double() if adaptive else ref * gain.double().view(1, -1, 1) + off.double().view(1, -1, 1)
    P = ops.dwconv_norm_planes(x.to(DEV), None if w is None else w.to(DEV), None if b is None else b.to(DEV),
                               None if adaptive else gain.to(DEV), None if adaptive else off.to(DEV),
                               cond.to(DEV) if adaptive else None, 5, 5 + c, planes=planes)
    got = ops.planes_to_float(P, n, c, t, planes)
    assert relerr(got, ref) < (1e-5 if planes == 2 else 2e-6)


@pytest.mark.parametrize("c,l", [(64, 36000), (-64, 36000), (16, 72000), (8, 144000)])
def test_fused_filter_blocks_are_deterministic_at_batch_scale(c, l):
    """128 windows (the bench's window batch), 100 launches on the same inputs: every output bitwise the first one
    (the scheduling fence of filter_block64_kernel was guarded by 3 repetitions on tiny shapes only).  c = -64: the
    64-channel block with its k5 convs on one fp16 plane (alive_filter_block64_range_fp16, the default of decoder precision mode 1)."""
    plain, c = c < 0, abs(c)
    from module import _native as nat
    N, lf = 128, 450
    L_ = nat.lib()
    gen = torch.Generator(device=DEV).manual_seed(3)
    film = torch.randn(N, 4128, lf, device=DEV, generator=gen)
    x = torch.randn(N, c, l, device=DEV, generator=gen)
    skip = torch.randn(N, c, l, device=DEV, generator=gen)
    out = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    if c == 64:
        w = (torch.randn(L_.alive_filter_block64_weights(), device=DEV, generator=gen) * 0.05).to(torch.bfloat16)
        b = torch.randn(7, 64, device=DEV, generator=gen) * 0.1

        def run():
            if plain:
                nat.check(L_.alive_filter_block64_range_fp16(x.data_ptr(), N, l, w.data_ptr(), b.data_ptr(), film.data_ptr(), 4128, lf, 3072,
                                                             0, 0, lf, skip.data_ptr(), out.data_ptr(), st))
            else:
                nat.check(L_.alive_filter_block64(x.data_ptr(), N, l, w.data_ptr(), b.data_ptr(), film.data_ptr(), 4128, lf, 3072,
                                                  skip.data_ptr(), out.data_ptr(), st))
    else:
        nw = L_.alive_filter_block_small_weights(c)
        w = (torch.cat([torch.randn(224, device=DEV, generator=gen) * 0.1,        # fp32 biases [7][32], then bf16 weight pairs in fp32 words
                       (torch.randn(2 * (nw - 224), device=DEV, generator=gen) * 0.1).to(torch.bfloat16).view(torch.int16).view(torch.float32)]).contiguous())

        def run():
            nat.check(L_.alive_filter_block_small(x.data_ptr(), N, c, l, w.data_ptr(), film.data_ptr(), 4128, lf, 100,
                                                  skip.data_ptr(), out.data_ptr(), st))
    first = None
    for rep in range(100):
        out.zero_()
        run()
        if first is None:
            first = out.clone()
            assert torch.isfinite(first).all()
        else:
            assert torch.equal(out, first), f"launch {rep} differs from the first"


@pytest.mark.parametrize("c", [64, -64, 16, 8])
def test_guarded_mfma_kernels_soak_under_a_concurrent_matrix_load(c):
    """Round 6: the accumulation-chain hazard of DESIGN 3.2b' has no root cause -- the stand-alone probe does not
    reproduce it -- so the conservative fence (ALIVE_CHAIN_GAP, enforced on the listings at build time) gets a second, independent
    check at run time: every fused FilterBlock that rests on it runs 1000 launches at a window batch that fills a fraction of the chip
    (24 windows) and 400 at the bench's (128), while a side stream keeps the matrix pipes and the power budget busy with large bf16
    GEMMs (the clock the chip holds under load moves by 20 %: MI355X_MICROARCH.md, DVFS give-back) -- every launch bitwise the first.
    Mismatches are counted on the device; one synchronisation per shape."""
    plain, c = c < 0, abs(c)
    from module import _native as nat
    L_ = nat.lib()
    lf, l = 450, {64: 36000, 16: 72000, 8: 144000}[c]
    side = torch.cuda.Stream()
    ga = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    gb = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    gc_ = torch.empty(8192, 8192, device=DEV, dtype=torch.bfloat16)
    for N, reps in ((24, 1000), (128, 400)):
        gen = torch.Generator(device=DEV).manual_seed(3 + N)
        film = torch.randn(N, 4128, lf, device=DEV, generator=gen)
        x = torch.randn(N, c, l, device=DEV, generator=gen)
        skip = torch.randn(N, c, l, device=DEV, generator=gen)
        out = torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream
        if c == 64:
            w = (torch.randn(L_.alive_filter_block64_weights(), device=DEV, generator=gen) * 0.05).to(torch.bfloat16)
            b = torch.randn(7, 64, device=DEV, generator=gen) * 0.1
            fn = L_.alive_filter_block64_range_fp16 if plain else L_.alive_filter_block64_range

            def run():
                nat.check(fn(x.data_ptr(), N, l, w.data_ptr(), b.data_ptr(), film.data_ptr(), 4128, lf, 3072, 0, 0, lf, skip.data_ptr(),
                             out.data_ptr(), st))
        else:
            nw = L_.alive_filter_block_small_weights(c)
            w = (torch.cat([torch.randn(224, device=DEV, generator=gen) * 0.1,
                           (torch.randn(2 * (nw - 224), device=DEV, generator=gen) * 0.1).to(torch.bfloat16).view(torch.int16).view(torch.float32)]).contiguous())

            def run():
                nat.check(L_.alive_filter_block_small(x.data_ptr(), N, c, l, w.data_ptr(), film.data_ptr(), 4128, lf, 100, skip.data_ptr(),
                                                      out.data_ptr(), st))
        run()
        first = out.clone()
        assert torch.isfinite(first).all()
        bad = torch.zeros((), dtype=torch.int64, device=DEV)
        side.wait_stream(torch.cuda.current_stream())
        for rep in range(reps):
            if rep % 2 == 0:                                       # (one 1.1-TFLOP GEMM ~ 1 ms: a steady load beside every launch)
                with torch.cuda.stream(side):
                    torch.mm(ga, gb, out=gc_)
            out.zero_()
            run()
            bad += (out != first).any()
        torch.cuda.synchronize()
        assert int(bad.item()) == 0, f"C = {c}{' (fp16 form)' if plain else ''}, {N} windows: {int(bad.item())} of {reps} launches differ from the first"
        del film, x, skip, out, first


@pytest.mark.parametrize("c,l", [(256, 4500), (64, 36000)])
def test_fused_sweep_blocks_soak_under_a_concurrent_matrix_load(c, l):
    """the sweep kernel of the 256- and 64-channel FilterBlock (csrc/filter_big.hip) under the same perturbation as the guarded kernels
    above: x 300 at 24 windows (blocks of few tiles, many warm-up tiles) and x 150 at the bench's 192, a bf16 GEMM load beside every
    launch, every launch bitwise the first.  (Its hand-overs: contexts through an L2 workspace between a block's tiles, LDS regions with
    three tenants per conv, two waves per SIMD behind block barriers.)"""
    import ctypes
    from module import _native as nat
    L_ = nat.lib()
    lf = 450
    entry, query = ((L_.alive_filter_block256_fp16, L_.alive_filter_block256_workspace_bytes) if c == 256 else
                    (L_.alive_filter_block64s_fp16, L_.alive_filter_block64s_workspace_bytes))
    side = torch.cuda.Stream()
    ga = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    gb = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    gc_ = torch.empty(8192, 8192, device=DEV, dtype=torch.bfloat16)
    for N, reps in ((24, 300), (192, 150)):
        gen = torch.Generator(device=DEV).manual_seed(5 + N)
        film = 0.05 * torch.randn(N, 4128, lf, device=DEV, generator=gen)
        x = 0.3 * torch.randn(N, c, l, device=DEV, generator=gen)
        skip = 0.3 * torch.randn(N, c, l, device=DEV, generator=gen)
        out = torch.empty_like(x)
        ws = [(torch.randn(c * 5 * c, device=DEV, generator=gen) * (0.32 / c ** 0.5)).to(torch.float16) for _ in range(6)]
        bs = [torch.randn(c, device=DEV, generator=gen) * 0.1 for _ in range(6)]
        W = (ctypes.c_void_p * 6)(*[w.data_ptr() for w in ws])
        B = (ctypes.c_void_p * 6)(*[b.data_ptr() for b in bs])
        nbytes = query(N, l)
        wsp = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        st = torch.cuda.current_stream().cuda_stream

        def run():
            nat.check(entry(x.data_ptr(), N, l, W, B, film.data_ptr(), 4128, lf, 0 if c == 256 else 3072, 0, 0, lf, skip.data_ptr(), out.data_ptr(),
                            wsp.data_ptr(), nbytes, st))
        run()
        first = out.clone()
        assert torch.isfinite(first).all()
        bad = torch.zeros((), dtype=torch.int64, device=DEV)
        side.wait_stream(torch.cuda.current_stream())
        for rep in range(reps):
            if rep % 2 == 0:
                with torch.cuda.stream(side):
                    torch.mm(ga, gb, out=gc_)
            out.zero_()
            wsp.random_(0, 255)                                     # (the workspace's old contents are nobody's input)
            run()
            bad += (out != first).any()
        torch.cuda.synchronize()
        assert int(bad.item()) == 0, f"C = {c}, {N} windows: {int(bad.item())} of {reps} launches differ from the first"
        del film, x, skip, out, first, wsp


def test_operators_write_only_their_outputs(monkeypatch):
    """every tensor an operator wrapper allocates for the C ABI (module/ops.py: outputs and plane buffers) is placed between
    two guard bands; after a battery of ragged shapes through alive_conv1d (fp32 / split / transposed / strided / skinny),
    alive_gemm_planes, alive_to_planes, alive_dwconv_norm(_planes), alive_channel_norm, alive_argmax_channels, alive_oscillator,
    the fused FilterBlocks and the filter edges, every band must be untouched"""
    from module import ops
    bands = []
    real_empty = torch.empty

    def guarded_empty(*shape, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        dev = kw.get("device", None)
        if dev is None or not str(dev).startswith("cuda"):
            return real_empty(*shape, **kw)
        dtype = kw.get("dtype", torch.float32)
        numel = 1
        for d in shape:
            numel *= int(d)
        esz = torch.empty(0, dtype=dtype).element_size()
        pad = 4096 // esz                                   # 4 KB either side keeps the 16-byte alignment of the payload
        buf = torch.zeros(numel + 2 * pad, dtype=dtype, device=dev)
        raw = buf.view(torch.uint8)
        raw[:4096] = 0x5A
        raw[raw.numel() - 4096:] = 0x5A
        bands.append(raw)
        return buf[pad:pad + numel].view(*shape)
    monkeypatch.setattr(ops.torch, "empty", guarded_empty)

    def t(name, shape, scale=1.0):
        return g(name, shape, scale=scale).to(DEV)

    # conv1d: exact fp32, split bf16 (2 / 3 planes), ragged Co / Ci / T, causal reflect padding with the FiLM second output
    for prec in ("fp32", "bf16x3", "bf16x6"):
        ops.conv1d(t("g1", (2, 37, 131)), t("g1w", (45, 37, 5), 0.1), t("g1b", (45,)), dilation=2, pad_left=8, pad_mode=1, out_len=131,
                   precision=prec)
        ops.conv1d(t("g2", (3, 64, 257)), t("g2w", (64, 64, 1), 0.1), t("g2b", (64,)), act="gelu", precision=prec)
    film, _ = ops.conv1d(t("gc", (2, 24, 19)), t("gcw", (128, 24, 1), 0.1), t("gcb", (128,)))
    ops.conv1d(t("g3", (2, 64, 190)), t("g3w", (64, 64, 5), 0.1), t("g3b", (64,)), dilation=1, pad_left=4, pad_mode=1, out_len=190,
               residual=t("g3r", (2, 64, 190)), film=film, film_scale_row=0, film_shift_row=64, precision="bf16x3")
    ops.conv1d(t("g4", (2, 16, 301)), t("g4w", (16, 8, 2), 0.1), t("g4b", (8,)), transposed=True)                    # ConvTranspose r = 2
    ops.conv1d(t("g5", (2, 64, 45)), t("g5w", (64, 16, 8), 0.1), t("g5b", (16,)), transposed=True, precision="bf16x3")   # r = 8
    ops.conv1d(t("g6", (2, 16, 800)), t("g6w", (64, 16, 8), 0.1), t("g6b", (64,)), stride=8)                          # strided down
    ops.conv1d(t("g7", (1, 256, 9)), t("g7w", (256, 256, 5), 0.05), t("g7b", (256,)), pad_left=4, pad_mode=1, out_len=9,
               precision="bf16x3")                                                                                     # skinny (streaming)
    # plane GEMMs, plane conversion, norms, argmax
    for planes in (2, 3):
        x = t("g8", (3, 77, 131))
        P = ops.to_planes(x, planes)
        ops.gemm_planes(P, 3, 131, t("g8w", (150, 77, 1), 0.1), t("g8b", (150,)), planes=planes, act="gelu", want_planes=True)
        if planes == 3:                                     # the argmax epilogue exists for the 3-plane classifier GEMM
            ops.gemm_planes_argmax(P, 3, 131, t("g8a", (300, 77, 1), 0.1), t("g8c", (300,)), planes=planes)
        ops.dwconv_norm_planes(t("g9", (2, 64, 101)), t("g9w", (64, 1, 7), 0.3), t("g9b", (64,)), t("g9g", (64,)) + 1.0, t("g9o", (64,)),
                               planes=planes)
    ops.dwconv_norm(t("ga", (2, 48, 33)), t("gaw", (48, 1, 7), 0.3), t("gab", (48,)), gain=t("gag", (1, 48, 1)), offset=t("gao", (1, 48, 1)))
    ops.dwconv_norm(t("ga", (2, 48, 7)), t("gaw", (48, 1, 7), 0.3), t("gab", (48,)), gain=t("gag", (1, 48, 1)), offset=t("gao", (1, 48, 1)))
    ops.channel_norm(t("gb", (2, 256, 45)), t("gbg", (1, 256, 1)), t("gbo", (1, 256, 1)))
    ops.argmax_channels(t("gd", (2, 4096, 13)))
    # oscillator, filter edges
    amps = t("ge", (2, 64, 11)).abs()
    ops.oscillator(amps, torch.full((2, 1, 11), 140.0, device=DEV), phi_col=11 * 320 - 1)
    ops.filter_source_in(t("gf", (2, 1, 2000)), t("gfw", (8, 1, 7), 0.2), t("gfb", (8,)), t("gfd", (16, 8, 2), 0.2), t("gfe", (16,)))
    ops.filter_source_out(t("gh", (2, 8, 2000)), t("ghw", (1, 8, 7), 0.2), t("ghb", (1,)))
    torch.cuda.synchronize()
    assert len(bands) >= 25
    for raw in bands:
        assert bool((raw[:4096] == 0x5A).all() and (raw[raw.numel() - 4096:] == 0x5A).all()), "an operator wrote outside its output"


# ---- split-bf16 conv with plane-packed operands (round 3): LDS-DMA staging, plane second output ----------------
@pytest.mark.parametrize("kw,dil,t,with_res", [(5, 1, 900, True), (5, 2, 520, False), (5, 4, 4500, True), (1, 1, 300, False)])
def test_conv1d_split_plane_operands_equal_the_fp32_staged_kernel(kw, dil, t, with_res):
    """AliveConv.Xp / Zp (the decoder's 256-channel FilterBlock chains its convs through them): the input arrives already
    split into two bf16 planes (k-blocked, rows = time) and is staged by LDS-DMA; the gelu + FiLM second output leaves in the same
    format.  Same split (round to nearest even), same MFMA order: Y bitwise equal to the fp32-staged kernel, Zp bitwise the
    plane image of its Z.  First tile reflects at t = 0 (causal conv), last tile is ragged."""
    from module import ops
    n, c, lf = 3, 256, max(5, t // 10)
    x = g(f"px{kw}{dil}{t}", (n, c, t))
    w = g(f"pw{kw}{dil}{t}", (c, c, kw), scale=1.0 / np.sqrt(c * kw))
    b = g(f"pb{kw}{dil}{t}", (c,), scale=0.1)
    film = g(f"pf{kw}{dil}{t}", (n, 2 * c, lf))
    res = g(f"pr{kw}{dil}{t}", (n, c, t)) if with_res else None
    kwargs = dict(dilation=dil, pad_left=(kw - 1) * dil, pad_mode=1, out_len=t, residual=None if res is None else res.to(DEV),
                  film=film.to(DEV), film_scale_row=0, film_shift_row=c, precision="bf16x3")
    y0, z0 = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), **kwargs)
    xp = ops.to_planes(x.to(DEV), 2)
    y1, zp1 = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), x_planes=xp, z_planes=True, **kwargs)
    assert torch.equal(y0, y1)
    want = ops.to_planes(z0, 2)
    cols, cols_pad = n * t, (n * t + 127) // 128 * 128
    a = zp1.view(torch.bfloat16).view(2, c // 32, cols_pad, 32)[:, :, :cols]           # k-blocked planes (csrc/planes_layout.h)
    e = want.view(torch.bfloat16).view(2, c // 32, cols_pad, 32)[:, :, :cols]
    assert torch.equal(a.contiguous().view(torch.int16), e.contiguous().view(torch.int16))
    # plane input alone (fp32 second output) and plane output alone (fp32 input)
    y2, z2 = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), x_planes=xp, **kwargs)
    assert torch.equal(y2, y0) and torch.equal(z2, z0)
    y3, zp3 = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), z_planes=True, **kwargs)
    assert torch.equal(y3, y0) and torch.equal(zp3.view(torch.bfloat16).view(2, c // 32, cols_pad, 32)[:, :, :cols].contiguous().view(torch.int16),
                                            e.contiguous().view(torch.int16))


@pytest.mark.parametrize("n,ci,co,r,t", [(3, 16, 64, 2, 4500), (1, 16, 64, 2, 260), (2, 8, 40, 4, 1000), (2, 16, 20, 2, 130)])
def test_conv1d_exact_kernel_writes_its_output_as_planes_too(n, ci, co, r, t):
    """AliveConv.Yp (round 5): the decoder's downs[1] (decoder.py:186-188, 16 -> 64 channels, stride 2) leaves the planes downs[2]'s
    GEMM reads beside its fp32 skip tensor.  Y bitwise the conv without Yp; the planes bitwise alive_to_planes(Y) in the columns and
    channels that exist (ragged column tile, Co that is not a multiple of 32 or 16)."""
    from module import ops
    x = g(f"ypx{n}{ci}{t}", (n, ci, t)).to(DEV)
    w = g(f"ypw{ci}{co}{r}", (co, ci, r), scale=1.0 / np.sqrt(ci * r)).to(DEV)
    b = g(f"ypb{co}", (co,), scale=0.1).to(DEV)
    y0, _ = ops.conv1d(x, w, b, stride=r)
    y1, yp = ops.conv1d(x, w, b, stride=r, y_planes=True)
    assert torch.equal(y0, y1)
    want = ops.to_planes(y0, 2)
    tout = t // r
    cols, cols_pad, cp = n * tout, (n * tout + 127) // 128 * 128, (co + 31) // 32 * 32
    a = yp.view(torch.bfloat16).view(2, cp // 32, cols_pad, 32)[:, :, :cols].permute(0, 2, 1, 3).reshape(2, cols, cp)[:, :, :co]
    e = want.view(torch.bfloat16).view(2, cp // 32, cols_pad, 32)[:, :, :cols].permute(0, 2, 1, 3).reshape(2, cols, cp)[:, :, :co]
    assert torch.equal(a.contiguous().view(torch.int16), e.contiguous().view(torch.int16))
    y2, yh = ops.conv1d(x, w, b, stride=r, y_planes=1)               # AliveConv.yp_planes = 1: ONE fp16 plane = alive_to_planes(Y, 1)
    assert torch.equal(y0, y2)
    want1 = ops.to_planes(y0, 1)
    a1 = yh.view(torch.int16).view(1, cp // 32, cols_pad, 32)[:, :, :cols].permute(0, 2, 1, 3).reshape(1, cols, cp)[:, :, :co]
    e1 = want1.view(torch.int16).view(1, cp // 32, cols_pad, 32)[:, :, :cols].permute(0, 2, 1, 3).reshape(1, cols, cp)[:, :, :co]
    assert torch.equal(a1.contiguous(), e1.contiguous())
    with pytest.raises(Exception, match="Yp"):
        ops.conv1d(x, w, b, stride=r, y_planes=True, act="gelu")


@pytest.mark.parametrize("co,ci,kw,dil,n,t,planes_io", [(256, 256, 5, 2, 3, 900, True), (256, 256, 5, 4, 2, 4500, True), (256, 256, 1, 1, 2, 520, False),
                                                        (192, 96, 5, 1, 2, 333, False), (40, 64, 3, 1, 2, 257, False)])
def test_conv1d_plain_fp16_is_the_float64_conv_of_the_rounded_operands(co, ci, kw, dil, n, t, planes_io):
    """AliveConv.precision 3 (round 5; the decoder's 256-channel FilterBlock, decoder.py:128-134): one MFMA per product on operands
    rounded to fp16 (nearest even), fp32 accumulate.  Against float64 on the SAME rounded operands only the accumulation differs
    (tolerance 2e-5 of the output scale); against the unrounded conv the difference is fp16's 2^-12 per operand.  With plane operands:
    Y bitwise the fp32-staged kernel's, the one-plane Zp bitwise alive_to_planes(Z, 1) = fp16(Z)."""
    from module import ops
    x = g(f"bx{co}{kw}{dil}{t}", (n, ci, t))
    w = g(f"bw{co}{kw}{dil}{t}", (co, ci, kw), scale=1.0 / np.sqrt(ci * kw))
    b = g(f"bb{co}{kw}{dil}{t}", (co,), scale=0.1)
    res = g(f"br{co}{kw}{dil}{t}", (n, co, t))
    kwargs = dict(dilation=dil, pad_left=(kw - 1) * dil, pad_mode=1, out_len=t, residual=res.to(DEV), precision="fp16")
    y, _ = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), **kwargs)
    xr, wr = x.half().double(), w.half().double()
    xp = torch.nn.functional.pad(xr, ((kw - 1) * dil, 0), mode="reflect") if kw > 1 else xr
    want = torch.nn.functional.conv1d(xp, wr, b.double(), dilation=dil) + res.double()
    assert (y.cpu().double() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    exact = torch.nn.functional.conv1d(torch.nn.functional.pad(x.double(), ((kw - 1) * dil, 0), mode="reflect") if kw > 1 else x.double(),
                                       w.double(), b.double(), dilation=dil) + res.double()
    err = (y.cpu().double() - exact).pow(2).mean().sqrt().item()
    assert 2e-6 < err < 8e-4, err                                   # really one fp16 plane (2^-12 per operand): not split bf16, not plain bf16
    if planes_io:
        lf = max(5, t // 10)
        film = g(f"bf{co}{kw}{dil}{t}", (n, 2 * co, lf)).to(DEV)
        kw2 = dict(kwargs, film=film, film_scale_row=0, film_shift_row=co)
        y0, z0 = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), **kw2)
        assert torch.equal(y0, y)
        x1 = ops.to_planes(x.to(DEV), 1)
        y1, zp = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), x_planes=x1, z_planes=True, **kw2)
        assert torch.equal(y1, y)
        cols, cols_pad = n * t, (n * t + 127) // 128 * 128
        e = ops.to_planes(z0, 1).view(torch.int16).view(1, co // 32, cols_pad, 32)[0, :, :cols]
        a = zp.view(torch.int16).view(1, co // 32, cols_pad, 32)[0, :, :cols]
        assert torch.equal(a.contiguous(), e.contiguous())
    if n * t > 96:
        with pytest.raises(Exception, match="batch form"):          # no few-column (streaming) form of the plain kernel
            ops.conv1d(x[:1, :, :40].to(DEV), w.to(DEV), b.to(DEV), dilation=dil, pad_left=(kw - 1) * dil, pad_mode=1, out_len=40, precision="fp16")


@pytest.mark.parametrize("n,c,t", [(2, 641, 450), (1, 512, 37), (3, 40, 130)])
def test_to_planes_one_plane_is_fp16(n, c, t):
    """alive_to_planes(planes = 1): ONE fp16 plane, round to nearest even, saturated at +-65504, zero padded like the bf16 planes."""
    from module import ops
    x = g(f"tq{n}{c}{t}", (n, c, t))
    x[0, 0, 0], x[0, 1, 0], x[0, 2, 0] = 1.0e6, -7.0e4, 3.0e-6      # saturates / a subnormal
    P = ops.to_planes(x.to(DEV), 1)
    cp, cols_pad = (c + 31) // 32 * 32, (n * t + 127) // 128 * 128
    raw = P.view(torch.float16).view(cp // 32, cols_pad, 32).permute(1, 0, 2).reshape(cols_pad, cp)
    want = x.clamp(-65504.0, 65504.0).half().permute(0, 2, 1).reshape(n * t, c)
    assert torch.equal(raw[:n * t, :c].cpu().view(torch.int16), want.view(torch.int16))
    assert raw[n * t:].float().abs().sum().item() == 0.0 and raw[:, c:].float().abs().sum().item() == 0.0


@pytest.mark.parametrize("co,ci,n,t,act", [(1536, 512, 2, 450, "gelu"), (512, 1536, 3, 37, None), (64, 512, 2, 24, None),
                                           (200, 96, 2, 130, "gelu"), (256, 2560, 2, 450, None)])     # K = 96: the 32-deep kernel; the others 64-deep steps
def test_gemm_planes_one_plane_is_the_float64_product_of_the_fp16_operands(co, ci, n, t, act):
    """AliveGemm.planes = 1 (round 5; the pointwise convs of the decoder's ConvNeXt layers, common.py:74-82): plain fp16 operands (one
    plane each), fp32 accumulate; the plane output is ONE fp16 plane = alive_to_planes(Y, 1)."""
    from module import ops
    x = g(f"g1x{co}{ci}{t}", (n, ci, t)); w = g(f"g1w{co}{ci}{t}", (co, ci, 1), scale=1.0 / np.sqrt(ci)); b = g(f"g1b{co}", (co,), scale=0.1)
    P = ops.to_planes(x.to(DEV), 1)
    y, pout = ops.gemm_planes(P, n, t, w.to(DEV), b.to(DEV), planes=1, act=act, want_planes=True)
    want = torch.einsum("oc,nct->not", w[:, :, 0].half().double(), x.half().double()) + b.double().view(1, -1, 1)
    if act == "gelu":
        want = torch.nn.functional.gelu(want)
    assert (y.cpu().double() - want).abs().max().item() <= 3e-5 * max(1.0, want.abs().max().item())
    cols, cols_pad, cp = n * t, (n * t + 127) // 128 * 128, (co + 31) // 32 * 32
    e = ops.to_planes(y, 1).view(torch.int16).view(1, cp // 32, cols_pad, 32)[0, :, :cols]
    a = pout.view(torch.int16).view(1, cp // 32, cols_pad, 32)[0, :, :cols]
    assert torch.equal(a.contiguous(), e.contiguous())


@pytest.mark.parametrize("co,ci,n,t,act", [(1536, 512, 2, 450, "gelu"), (512, 1536, 3, 37, None), (200, 96, 2, 130, None)])
def test_gemm_planes_fp16_split_is_fp32_grade(co, ci, n, t, act):
    """AliveGemm.f16s (round 5; the pointwise convs of the encoders' ConvNeXt layers, common.py:54-62): both operands as two fp16 planes
    of a power-of-two multiple of the values, three MFMAs per product.  Against float64: the error of the three-plane bf16 form
    (six MFMAs) within a factor of a few -- fp32-grade -- and two orders below the two-plane bf16 form; the plane output in the same
    format reproduces the fp32 output to 22 bits."""
    import ctypes as C
    from module import ops, _native as nat
    from module._pack import pack_conv_split_f16s
    x = g(f"fsx{co}{ci}{t}", (n, ci, t)); w = g(f"fsw{co}{ci}{t}", (co, ci, 1), scale=1.0 / np.sqrt(ci)); b = g(f"fsb{co}", (co,), scale=0.1)
    x[0, 0, 0], x[0, 1, 1] = 3.0e-6, 200.0                          # a tiny and a large activation
    want = torch.einsum("oc,nct->not", w[:, :, 0].double(), x.double()) + b.double().view(1, -1, 1)
    if act == "gelu":
        want = torch.nn.functional.gelu(want)
    scale = 256.0
    L = nat.lib()
    xd = x.to(DEV)
    # the producer of the format: dw conv off, norm off is not available -- build the planes from alive_to_planes's arithmetic on the host
    cp, cols, cols_pad = (ci + 31) // 32 * 32, n * t, (n * t + 127) // 128 * 128
    flat = torch.zeros(cols_pad, cp, dtype=torch.float32)
    flat[:cols, :ci] = x.permute(0, 2, 1).reshape(cols, ci) * scale
    hi = flat.clamp(-65504, 65504).half(); lo = (flat - hi.float()).clamp(-65504, 65504).half()
    P = torch.stack([hi, lo], 0).view(2, cols_pad, cp // 32, 32).permute(0, 2, 1, 3).contiguous().to(DEV)
    W5, ws = pack_conv_split_f16s(w.to(DEV))
    bd = b.to(DEV)
    y = torch.empty(n, co, t, device=DEV)
    cop = (co + 31) // 32 * 32
    pout = torch.zeros(2, cop // 32, cols_pad, 32, dtype=torch.float16, device=DEV)
    d = nat.AliveGemm()
    d.W, d.bias, d.P = W5[3:].contiguous().data_ptr(), bd.data_ptr(), P.data_ptr()
    keep = W5[3:].contiguous(); d.W = keep.data_ptr()
    d.N, d.T, d.Ci, d.Co, d.planes, d.act = n, t, ci, co, 2, ops.ACT[act]
    d.Y, d.Pout = y.data_ptr(), pout.data_ptr()
    d.f16s, d.wscale, d.in_unscale, d.pout_scale = 1, ws.data_ptr(), 1.0 / scale, scale
    nat.check(L.alive_gemm_planes(C.byref(d), nat.stream()), "alive_gemm_planes")
    ref_scale = max(1.0, want.abs().max().item())
    err = (y.cpu().double() - want).abs().max().item()
    y3, _ = ops.gemm_planes(ops.to_planes(xd, 3), n, t, w.to(DEV), bd, planes=3, act=act)
    y2, _ = ops.gemm_planes(ops.to_planes(xd, 2), n, t, w.to(DEV), bd, planes=2, act=act)
    e3, e2 = (y3.cpu().double() - want).abs().max().item(), (y2.cpu().double() - want).abs().max().item()
    # fp32-grade = what the three-plane bf16 form reaches on the same data (with K = 1536 and an activation of 200 both sit on the fp32
    # accumulation, 2.2e-5); the two-plane bf16 form is 3 - 60 x worse
    assert err <= 4e-6 * ref_scale and err <= 2 * e3 + 1e-7 * ref_scale and err < 0.5 * e2, (err, e3, e2)
    back = (pout[0].float() + pout[1].float()).permute(1, 0, 2).reshape(cols_pad, cop)[:cols, :co] / scale
    back = back.view(n, t, co).permute(0, 2, 1)
    assert (back.cpu().double() - y.cpu().double()).abs().max().item() <= 2.0 ** -20 * ref_scale


def test_dwconv_norm_planes_fp16_split():
    """alive_dwconv_norm_planes_f16s: the two fp16 planes of 256 x the normalised tensor reproduce alive_dwconv_norm's fp32 output to
    22 bits of its largest element; padding is zero."""
    from module import ops, _native as nat
    n, c, t = 2, 512, 333
    x = g("fsn", (n, c, t)).to(DEV)
    dw_w, dw_b = g("fsnw", (c, 1, 7), scale=0.3).to(DEV), g("fsnb", (c,), scale=0.1).to(DEV)
    gain, off = (1.0 + g("fsng", (c,), scale=0.1)).to(DEV), g("fsno", (c,), scale=0.1).to(DEV)
    y = ops.dwconv_norm(x, dw_w, dw_b, gain=gain, offset=off)
    cols, cols_pad = n * t, (n * t + 127) // 128 * 128
    P = torch.zeros(2, c // 32, cols_pad, 32, dtype=torch.float16, device=DEV)
    fl = lambda v: nat.ptr(v.reshape(-1).contiguous())
    w7, keep = dw_w.reshape(-1).contiguous(), [gain.contiguous(), off.contiguous()]
    nat.check(nat.lib().alive_dwconv_norm_planes_f16s(nat.ptr(x), n, c, t, nat.ptr(w7), nat.ptr(dw_b), 0, nat.ptr(keep[0]), nat.ptr(keep[1]),
                                                      None, 0, 0, 0, 1e-4, 256.0, nat.ptr(P), nat.stream()), "alive_dwconv_norm_planes_f16s")
    back = ((P[0].float() + P[1].float()) / 256.0).permute(1, 0, 2).reshape(cols_pad, c)
    assert back[cols:].abs().sum().item() == 0.0
    got = back[:cols].view(n, t, c).permute(0, 2, 1)
    assert (got - y).abs().max().item() <= 2.0 ** -21 * y.abs().max().item()


def test_fp16_saturations_are_counted_and_refused():
    """alive_f16_saturations: every value a producer of fp16 planes had to saturate is counted (one counter per kernel file), and
    Converter.check_fp16_range turns a non-zero count into an error -- an out-of-range activation never passes silently."""
    from module import ops
    from module.pipeline import Converter
    ops.f16_saturations(reset=True)
    x = g("sat", (2, 64, 130))
    ops.to_planes(x.to(DEV), 1)
    assert ops.f16_saturations() == 0
    x[1, 3, 7], x[0, 5, 9] = 7.0e4, -1.0e9
    ops.to_planes(x.to(DEV), 1)
    assert ops.f16_saturations() >= 2                     # (counted per converted pair)
    with pytest.raises(RuntimeError, match="fp16"):
        Converter.check_fp16_range()
    assert ops.f16_saturations() == 0                     # the check resets the counters
    Converter.check_fp16_range()


def test_fp16_saturation_count_ignores_padding_columns_of_a_gemm():
    """A plane buffer's padding columns (cols .. cols_pad - 1) hold whatever was there: the GEMM computes them (and skips them on the way
    out), but what it saturates THERE is not counted -- only real outputs can trip Converter.check_fp16_range."""
    from module import ops
    n, ci, co, t = 3, 512, 256, 37                                   # 111 columns in a 128-column buffer
    x = g("padx", (n, ci, t)); w = g("padw", (co, ci, 1), scale=1.0 / np.sqrt(ci)); b = g("padb", (co,), scale=0.1)
    P = ops.to_planes(x.to(DEV), 1)
    v = P.view(torch.int16).view(ci // 32, 128, 32)
    v[:, n * t:, :] = 0x7BFF                                           # 65504 in every padding element: their outputs saturate
    ops.f16_saturations(reset=True)
    y, pout = ops.gemm_planes(P, n, t, w.to(DEV), b.to(DEV), planes=1, want_planes=True)
    assert ops.f16_saturations() == 0
    want = torch.einsum("oc,nct->not", w[:, :, 0].half().double(), x.half().double()) + b.double().view(1, -1, 1)
    assert (y.cpu().double() - want).abs().max().item() <= 3e-5 * max(1.0, want.abs().max().item())
