"""The reference-captured block fixtures of tests/golden (oracle/gen_golden.py ran the REFERENCE modules on these inputs)
replayed through the exported operators of the C ABI -- alive_conv1d, alive_dwconv_norm, alive_channel_norm -- so that
every shared block of module/common.py:14-92 and module/decoder.py:13-48,105-195 is checked on the GPU against the
reference's own output, not only against the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = dict(rtol=3e-5, atol=3e-5)


def relerr(got, ref):
    """largest deviation relative to the RMS of the reference output (the fixtures run random weights: the U-Net's output
    reaches 1e13, so an absolute tolerance means nothing there)"""
    return ((got.double().cpu() - ref.double()).abs().max() / ref.double().pow(2).mean().sqrt()).item()


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    d = {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}
    sd = {k[3:]: v.to(DEV) for k, v in d.items() if k.startswith("w::")}
    return d, sd


def film_rows(ops, c, sd, prefixes):
    """[to_scale; to_shift] of every modulated conv as ONE 1x1 conv, `+1` on the scale rows (decoder.py:109-115)"""
    ws, bs, post = [], [], []
    for p in prefixes:
        n = sd[p + "to_scale.bias"].numel()
        ws += [sd[p + "to_scale.weight"], sd[p + "to_shift.weight"]]
        bs += [sd[p + "to_scale.bias"], sd[p + "to_shift.bias"]]
        post += [torch.ones(n, device=DEV), torch.zeros(n, device=DEV)]
    film, _ = ops.conv1d(c.to(DEV), torch.cat(ws, 0), torch.cat(bs, 0), post_add=torch.cat(post))
    return film


def test_channel_norm_fixture(golden_dir):
    from module import ops
    d, sd = load(golden_dir, "blk_channel_norm")
    y = ops.channel_norm(d["x"].to(DEV), sd["scale"], sd["shift"])
    torch.testing.assert_close(y.cpu(), d["y"], **TOL)


def test_adaptive_channel_norm_fixture(golden_dir):
    """common.py:29-41: scale / shift are 1x1 convs of the condition (no +1); the norm kernel takes them as rows of a
    condition tensor.  The depthwise stage of the fused kernel is set to the identity tap."""
    from module import ops
    d, sd = load(golden_dir, "blk_adaptive_channel_norm")
    c = d["x"].shape[1]
    cond, _ = ops.conv1d(d["c"].to(DEV), torch.cat([sd["scale.weight"], sd["shift.weight"]], 0),
                         torch.cat([sd["scale.bias"], sd["shift.bias"]], 0))
    ident = torch.zeros(c, 1, 7, device=DEV)
    ident[:, 0, 3] = 1.0
    y = ops.dwconv_norm(d["x"].to(DEV), ident, torch.zeros(c, device=DEV), cond=cond, scale_row=0, shift_row=c)
    torch.testing.assert_close(y.cpu(), d["y"], **TOL)


@pytest.mark.parametrize("adaptive", [False, True])
def test_convnext_fixture(golden_dir, adaptive):
    """ConvNeXt1d / AdaptiveConvNeXt1d (common.py:44-82): dw conv + norm -> pw1 + GELU -> pw2 * scale + x"""
    from module import ops
    d, sd = load(golden_dir, "blk_adaptive_convnext" if adaptive else "blk_convnext")
    x = d["x"].to(DEV)
    c = x.shape[1]
    if adaptive:
        cond, _ = ops.conv1d(d["c"].to(DEV), torch.cat([sd["norm.scale.weight"], sd["norm.shift.weight"]], 0),
                             torch.cat([sd["norm.scale.bias"], sd["norm.shift.bias"]], 0))
        y = ops.dwconv_norm(x, sd["dw_conv.weight"], sd["dw_conv.bias"], cond=cond, scale_row=0, shift_row=c)
    else:
        y = ops.dwconv_norm(x, sd["dw_conv.weight"], sd["dw_conv.bias"], gain=sd["norm.scale"], offset=sd["norm.shift"])
    y, _ = ops.conv1d(y, sd["pw_conv1.weight"], sd["pw_conv1.bias"], act="gelu")
    y, _ = ops.conv1d(y, sd["pw_conv2.weight"], sd["pw_conv2.bias"], ch_scale=sd["scale"].reshape(-1), residual=x)
    torch.testing.assert_close(y.cpu(), d["y"], **TOL)


@pytest.mark.parametrize("dil", [1, 2, 4])
def test_causal_conv_fixture(golden_dir, dil):
    """CausalConv1d (common.py:85-92): reflect pad (k - 1) d on the left only"""
    from module import ops
    d, sd = load(golden_dir, f"blk_causal_conv_d{dil}")
    x = d["x"].to(DEV)
    y, _ = ops.conv1d(x, sd["conv.weight"], sd["conv.bias"], dilation=dil, pad_left=4 * dil, pad_mode=1, out_len=x.shape[2])
    torch.testing.assert_close(y.cpu(), d["y"], **TOL)


def test_modulated_causal_conv_fixture(golden_dir):
    """ModulatedCausalConv1d (decoder.py:105-119) on its own has no GELU in front, so the fused GELU + FiLM epilogue does not
    apply: the FiLM rows come from alive_conv1d, the (reference-exact) linear interpolation from torch on the device, the
    conv from alive_conv1d."""
    from module import ops
    d, sd = load(golden_dir, "blk_modulated_causal_conv")
    x, dil = d["x"].to(DEV), int(d["dilation"])
    c = x.shape[1]
    film = film_rows(ops, d["c"], sd, [""])
    xm = x * F.interpolate(film[:, :c], x.shape[2], mode="linear") + F.interpolate(film[:, c:], x.shape[2], mode="linear")
    y, _ = ops.conv1d(xm, sd["conv.conv.weight"], sd["conv.conv.bias"], dilation=dil, pad_left=4 * dil, pad_mode=1,
                      out_len=x.shape[2])
    torch.testing.assert_close(y.cpu(), d["y"], **TOL)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_filter_res_block_fixture(golden_dir, precision):
    """FilterResBlock (decoder.py:122-134): GELU -> FiLM -> causal conv, twice, + residual -- through the dual-output
    epilogue (an identity 1x1 conv produces the first modulated input)"""
    from module import ops
    d, sd = load(golden_dir, "blk_filter_res_block")
    x, dil = d["x"].to(DEV), int(d["dilation"])
    c = x.shape[1]
    film = film_rows(ops, d["c"], sd, ["c1.", "c2."])
    eye = torch.eye(c, device=DEV).unsqueeze(2)
    _, z1 = ops.conv1d(x, eye, None, film=film, film_scale_row=0, film_shift_row=c, want_raw=False)
    kw = dict(dilation=dil, pad_left=4 * dil, pad_mode=1, out_len=x.shape[2], precision=precision)
    _, z2 = ops.conv1d(z1, sd["c1.conv.conv.weight"], sd["c1.conv.conv.bias"], film=film, film_scale_row=2 * c,
                       film_shift_row=3 * c, want_raw=False, **kw)
    y, _ = ops.conv1d(z2, sd["c2.conv.conv.weight"], sd["c2.conv.conv.bias"], residual=x, **kw)
    e = relerr(y, d["y"])           # fp32: fmaf chains + the fused A&S 7.1.26 GELU (1.5e-7); split-bf16: ~2^-16 per product
    assert e < (5e-6 if precision == "fp32" else 1e-4), e


def test_f0_encoder_fixture(golden_dir):
    """F0Encoder (decoder.py:13-24): 1 -> C 1x1 conv, sin, C -> C 1x1 conv; c1 has K = 1 and feeds sin, so it must be the
    single fma(w, x, b) ATen produces"""
    from module import ops
    d, sd = load(golden_dir, "blk_f0_encoder")
    y, _ = ops.conv1d(d["f0"].to(DEV), sd["c1.weight"], sd["c1.bias"], act="sin")
    y, _ = ops.conv1d(y, sd["c2.weight"], sd["c2.bias"])
    torch.testing.assert_close(y.cpu(), d["y"], rtol=1e-3, atol=1e-3)


def test_filter_fixture(golden_dir):
    """Filter.forward (decoder.py:153-195), the whole U-Net at the fixture's small widths [4, 8, 16, 32], conv by conv through
    alive_conv1d: source_in (k7, zero pad 3), strided downs, causal mid conv, transposed ups on (x + skip), a FilterBlock per
    scale through the dual-output epilogue, source_out"""
    from module import ops
    d, sd = load(golden_dir, "blk_filter")
    src, c = d["src"].to(DEV), d["c"]
    prefixes = [f"blocks.{i}.blocks.{j}.{cc}." for i in range(4) for j in range(3) for cc in ("c1", "c2")]
    film = film_rows(ops, c, sd, prefixes)
    row = {}
    r = 0
    for p in prefixes:
        n = sd[p + "to_scale.bias"].numel()
        row[p] = (r, r + n)
        r += 2 * n
    x, _ = ops.conv1d(src, sd["source_in.weight"], sd["source_in.bias"], pad_left=3, out_len=src.shape[2])
    skips = []
    for i in range(4):
        w = sd[f"downs.{i}.weight"]
        x, _ = ops.conv1d(x, w, sd[f"downs.{i}.bias"], stride=w.shape[2])
        skips.append(x)
    x, _ = ops.conv1d(x, sd["mid_conv.conv.weight"], sd["mid_conv.conv.bias"], pad_left=4, pad_mode=1, out_len=x.shape[2])
    for i, s in enumerate(reversed(skips)):
        x, _ = ops.conv1d(x + s, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], transposed=True)
        L = x.shape[2]
        b = f"blocks.{i}."
        s0, h0 = row[b + "blocks.0.c1."]
        hres, z = ops.conv1d(x, sd[b + "input_conv.weight"], sd[b + "input_conv.bias"], film=film, film_scale_row=s0, film_shift_row=h0)
        for j in range(3):
            dil = 2 ** j
            kw = dict(dilation=dil, pad_left=4 * dil, pad_mode=1, out_len=L)
            s2, h2 = row[b + f"blocks.{j}.c2."]
            _, z2 = ops.conv1d(z, sd[b + f"blocks.{j}.c1.conv.conv.weight"], sd[b + f"blocks.{j}.c1.conv.conv.bias"], film=film,
                               film_scale_row=s2, film_shift_row=h2, want_raw=False, **kw)
            nxt = row.get(b + f"blocks.{j + 1}.c1.")
            hres, z = ops.conv1d(z2, sd[b + f"blocks.{j}.c2.conv.conv.weight"], sd[b + f"blocks.{j}.c2.conv.conv.bias"], residual=hres,
                                 film=film if nxt else None, film_scale_row=nxt[0] if nxt else 0, film_shift_row=nxt[1] if nxt else 0, **kw)
        x = hres
    y, _ = ops.conv1d(x, sd["source_out.weight"], sd["source_out.bias"], pad_left=3, out_len=x.shape[2])
    e = relerr(y, d["y"])
    assert e < 5e-5, e
