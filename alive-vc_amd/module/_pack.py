"""Re-pack reference state_dicts into the GEMM-ready tables of libalive_vc.so.

One-time host/device plumbing at checkpoint-load time (torch ops, no kernels
of our own): every conv weight becomes A[Co_pad16][K_pad16] with K = ci*KW + j
(zero padded), transposed convs become rows (co, j), and the per-layer FiLM /
adaptive-norm projections that share an input are concatenated into one GEMM.
Table names are the ones `alive_weight_name()` reports.
"""
import torch


def _pad16(n):
    return (n + 15) // 16 * 16


def pack_matrix(w2d):
    """[rows, K] -> zero-padded [rows_pad16, K_pad16] fp32 contiguous."""
    r, k = w2d.shape
    out = torch.zeros(_pad16(r), _pad16(k), dtype=torch.float32, device=w2d.device)
    out[:r, :k] = w2d
    return out.contiguous()


def pack_conv(w):
    """Conv1d weight [Co, Ci, KW] -> [Co_pad, (Ci*KW)_pad]."""
    return pack_matrix(w.reshape(w.shape[0], -1).float())


def pack_convT(w, b):
    """ConvTranspose1d(k == stride) weight [Ci, Co, r] -> rows (co, j): [Co*r, Ci]; bias repeated per j."""
    ci, co, r = w.shape
    a = w.permute(1, 2, 0).reshape(co * r, ci).float()
    return pack_matrix(a), b.float().repeat_interleave(r).contiguous()


def split_bf16(w, planes=2):
    """fp32 -> `planes` bf16 tensors whose sum reproduces w to 8*planes mantissa bits (round-to-nearest each step)."""
    out, r = [], w.float()
    for _ in range(planes):
        h = r.bfloat16()
        out.append(h)
        r = r - h.float()
    return out


def _split_rows(w, planes=2):
    """Conv1d weight [Co, Ci, KW] -> bf16 [planes, Co_pad16, KW * Ci_pad32], row-major, tap-major k = j * Ci_pad + ci."""
    co, ci, kw = w.shape
    co_pad, ci_pad = _pad16(co), (ci + 31) // 32 * 32
    a = torch.zeros(co_pad, kw, ci_pad, dtype=torch.float32, device=w.device)
    a[:co, :, :ci] = w.float().permute(0, 2, 1)
    return torch.stack(split_bf16(a.reshape(co_pad, kw * ci_pad), planes), 0).contiguous()


def pack_conv_split(w, planes=2):
    """Conv1d weight [Co, Ci, KW] -> bf16 [planes, K / 32, Co_pad16, 32]: the K-BLOCKED plane format of csrc/planes_layout.h,
    K = KW * Ci_pad32, tap-major k = j * Ci_pad + ci; element (plane, row, k) at [plane, k // 32, row, k % 32].
    planes = 2: "bf16x3" (3 MFMAs per product, ~2^-16); planes = 3: "bf16x6" (6 MFMAs, ~2^-24, fp32-grade)."""
    a = _split_rows(w, planes)
    p, co_pad, k = a.shape
    return a.view(p, co_pad, k // 32, 32).permute(0, 2, 1, 3).contiguous()


def pack_conv_split_h(w):
    """pack_conv_split(w, 2) followed by a THIRD slab of the same shape: the weights as ONE fp16 plane (round to nearest even, saturated at
    +-65504), the operand of the plain kernels (AliveConv.precision 3, AliveGemm.planes 1; csrc/networks.hip::plain_w).  Stored in the same
    bf16-typed tensor: [3, K / 32, Co_pad16, 32], slab 2 holds fp16 bits.  The split kernels read slabs 0 and 1 only."""
    two = pack_conv_split(w, 2)
    co, ci, kw = w.shape
    co_pad, ci_pad = _pad16(co), (ci + 31) // 32 * 32
    a = torch.zeros(co_pad, kw, ci_pad, dtype=torch.float32, device=w.device)
    a[:co, :, :ci] = w.float().permute(0, 2, 1)
    h = a.reshape(co_pad, kw * ci_pad).clamp(-65504.0, 65504.0).to(torch.float16)
    h = h.view(co_pad, kw * ci_pad // 32, 32).permute(1, 0, 2).contiguous()
    return torch.cat([two.view(torch.int16), h.view(torch.int16).unsqueeze(0)], 0).view(torch.bfloat16).contiguous()


def pack_conv_split_f16s(w):
    """pack_conv_split(w, 3) followed by TWO fp16 slabs of the same shape -- hi = fp16(s w), lo = fp16(s w - hi), s = 2^e the largest power
    of two with s max|w| <= 16384 -- and the fp32 scalar 1 / s: the weight operand of alive_gemm_planes with AliveGemm.f16s (three MFMAs
    per product at 22 significand bits; csrc/networks.hip::convnext_layer).  -> ([5, K / 32, Co_pad16, 32] bf16-typed, [1] fp32)"""
    three = pack_conv_split(w, 3)
    co, ci, kw = w.shape
    co_pad, ci_pad = _pad16(co), (ci + 31) // 32 * 32
    a = torch.zeros(co_pad, kw, ci_pad, dtype=torch.float32, device=w.device)
    a[:co, :, :ci] = w.float().permute(0, 2, 1)
    a = a.reshape(co_pad, kw * ci_pad)
    m = float(a.abs().max())
    e = 0 if not (m > 0.0) or m != m or m == float("inf") else int(torch.floor(torch.log2(torch.tensor(16384.0 / m, dtype=torch.float64))))
    e = max(-24, min(24, e))
    sc = a * (2.0 ** e)                                                     # exact: a power of two
    hi = sc.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = (sc - hi.float()).clamp(-65504.0, 65504.0).to(torch.float16)
    kb = lambda t: t.view(co_pad, kw * ci_pad // 32, 32).permute(1, 0, 2).contiguous().view(torch.int16).unsqueeze(0)
    W = torch.cat([three.view(torch.int16), kb(hi), kb(lo)], 0).view(torch.bfloat16).contiguous()
    return W, torch.tensor([2.0 ** -e], dtype=torch.float32, device=w.device)


def unpack_conv_split(W):
    """k-blocked [planes, K / 32, Co_pad, 32] -> row-major [planes, Co_pad, K] (tests, tools)"""
    p, kb, co_pad, _ = W.shape
    return W.permute(0, 2, 1, 3).reshape(p, co_pad, kb * 32)


def pack_conv_split3(w):
    return pack_conv_split(w, 3)


def pack_convT_split(w, b):
    """ConvTranspose1d(k == stride) weight [Ci, Co, r] -> rows (co, j) as a split 1x1 conv; bias repeated per j."""
    ci, co, r = w.shape
    a = w.permute(1, 2, 0).reshape(co * r, ci, 1)
    return pack_conv_split(a), b.float().repeat_interleave(r).contiguous()


def pack_convT_split_h(w, b):
    """pack_convT_split with the fp16 slab of pack_conv_split_h behind the two bf16 planes"""
    ci, co, r = w.shape
    a = w.permute(1, 2, 0).reshape(co * r, ci, 1)
    return pack_conv_split_h(a), b.float().repeat_interleave(r).contiguous()


def _vec(t):
    return t.reshape(-1).float().contiguous()


def _convnext(out, sd, src, dst, adaptive, pc=pack_conv):
    out[dst + ".dw_w"] = _vec(sd[src + ".dw_conv.weight"])
    out[dst + ".dw_b"] = _vec(sd[src + ".dw_conv.bias"])
    if not adaptive:
        out[dst + ".norm_gain"] = _vec(sd[src + ".norm.scale"])
        out[dst + ".norm_offset"] = _vec(sd[src + ".norm.shift"])
    out[dst + ".pw1.W"] = pc(sd[src + ".pw_conv1.weight"])
    out[dst + ".pw1.b"] = _vec(sd[src + ".pw_conv1.bias"])
    out[dst + ".pw2.W"] = pc(sd[src + ".pw_conv2.weight"])
    out[dst + ".pw2.b"] = _vec(sd[src + ".pw_conv2.bias"])
    out[dst + ".scale"] = _vec(sd[src + ".scale"])
    if pc is pack_conv_split3:       # the encoders: + the fp16 split image of the pointwise weights and its scale (alive_encoder_precision 1)
        out[dst + ".pw1.W"], out[dst + ".pw1.ws"] = pack_conv_split_f16s(sd[src + ".pw_conv1.weight"])
        out[dst + ".pw2.W"], out[dst + ".pw2.ws"] = pack_conv_split_f16s(sd[src + ".pw_conv2.weight"])


def pack_content_encoder(sd):
    """encoders: 3-plane split ("bf16x6", fp32-grade) -- a top-k / an argmax sit downstream of these GEMMs"""
    out = {"input.W": pack_conv_split3(sd["input_layer.weight"]), "input.b": _vec(sd["input_layer.bias"])}
    for i in range(4):
        _convnext(out, sd, f"mid_layers.{i}", f"mid{i}", False, pack_conv_split3)
    out["output.W"] = pack_conv_split3(sd["output_layer.weight"])
    out["output.b"] = _vec(sd["output_layer.bias"])
    return out


def pack_f0_estimator(sd):
    out = pack_content_encoder(sd)
    out["last_norm.gain"] = _vec(sd["last_norm.scale"])
    out["last_norm.offset"] = _vec(sd["last_norm.shift"])
    out["output.W"], out["output.ws"] = pack_conv_split_f16s(sd["output_layer.weight"])       # the classifier: + its fp16 split image
    return out


FILTER_CH = [256, 64, 16, 8]
# how each filter scale runs (must match F_MODE in csrc/networks.hip): "split" = conv by conv on the split-bf16 MFMA
# kernel (C = 256), "mid" = whole FilterBlock fused on the split-bf16 MFMA (C = 64, csrc/filter_mid.hip),
# "small" = whole FilterBlock fused on the f32 MFMA (C = 16, 8, csrc/filter_small.hip)
FILTER_MODE = ["split", "mid", "small", "small"]
SPLIT_SCALE = [m != "small" for m in FILTER_MODE]      # ConvTranspose in front of the scale on the split kernel


def pack_filter_mid(sd, prefix):
    """FilterBlock weights for the fused 64-channel kernel (csrc/filter_mid.hip):
    -> (bf16-typed flat: input conv [2][64][64], then 6 x [2][64][k = j*64 + ci], then 6 x [64][k] holding fp16 bits; fp32 biases [7][64])"""
    ws = [_split_rows(sd[prefix + ".input_conv.weight"]).reshape(-1)]
    bs = [sd[prefix + ".input_conv.bias"].float()]
    hs = []                                  # the k5 convs once more as ONE fp16 plane each [64][k = j*64 + ci] (the plain form, H16)
    for j in range(3):
        for cc in ("c1", "c2"):
            w = sd[f"{prefix}.blocks.{j}.{cc}.conv.conv.weight"]
            ws.append(_split_rows(w).reshape(-1))
            bs.append(sd[f"{prefix}.blocks.{j}.{cc}.conv.conv.bias"].float())
            co, ci, kw = w.shape
            hs.append(w.float().permute(0, 2, 1).reshape(co, kw * ci).clamp(-65504.0, 65504.0).to(torch.float16).view(torch.bfloat16).reshape(-1))
    return torch.cat(ws + hs).contiguous(), torch.stack(bs, 0).contiguous()


def pack_filter_small(sd, prefix):
    """FilterBlock weights for the fused 8/16-channel kernel (csrc/filter_small.hip, round 4: split-bf16 on the 32x32x16 MFMA):
    one fp32 buffer = biases [7][32] (input conv first; rows >= C zero), then -- bf16 pairs viewed as fp32 -- the input conv
    [2 planes][32 rows][16] (k = ci) and per conv q = 0..5 [2 planes][32 rows][KP] with k = tap * C + ci, KP = 80 (C = 16) / 48 (C = 8);
    output rows >= C and the padding k are zero.  Planes: hi = bf16(w), lo = bf16(w - hi)."""
    w_in = sd[prefix + ".input_conv.weight"].float()                           # [co, ci, 1]
    c = w_in.shape[0]
    kp = 80 if c == 16 else 48
    dev = w_in.device

    def planes(m, kpad):                 # [co, K] fp32 -> bf16 [2][32][kpad]
        full = torch.zeros(32, kpad, dtype=torch.float32, device=dev)
        full[:m.shape[0], :m.shape[1]] = m
        hi = full.to(torch.bfloat16)
        lo = (full - hi.float()).to(torch.bfloat16)
        return torch.stack([hi, lo], 0).reshape(-1)

    mats, biases = [planes(w_in[:, :, 0], 16)], [sd[prefix + ".input_conv.bias"].float()]
    for j in range(3):
        for cc in ("c1", "c2"):
            w = sd[f"{prefix}.blocks.{j}.{cc}.conv.conv.weight"].float()         # [co, ci, 5]
            mats.append(planes(w.permute(0, 2, 1).reshape(c, 5 * c), kp))        # k = tap * C + ci
            biases.append(sd[f"{prefix}.blocks.{j}.{cc}.conv.conv.bias"].float())
    bvec = torch.zeros(len(biases), 32, dtype=torch.float32, device=dev)
    for i, b in enumerate(biases):
        bvec[i, :b.numel()] = b
    wbits = torch.cat(mats).contiguous().view(torch.int16).view(torch.float32)   # two bf16 per fp32 word, bit for bit
    return torch.cat([bvec.reshape(-1), wbits]).contiguous()


def pack_decoder(sd):
    out = {}
    fe = "feature_extractor"
    out["fe.input.W"] = pack_conv_split(sd[fe + ".input_layer.weight"])
    out["fe.input.b"] = _vec(sd[fe + ".input_layer.bias"])
    out["fe.f0c1.W"] = pack_conv(sd[fe + ".f0_enc.c1.weight"])
    out["fe.f0c1.b"] = _vec(sd[fe + ".f0_enc.c1.bias"])
    out["fe.f0c2.W"] = pack_conv_split(sd[fe + ".f0_enc.c2.weight"])
    out["fe.f0c2.b"] = _vec(sd[fe + ".f0_enc.c2.bias"])
    ws, bs = [], []
    for i in range(4):
        p = f"{fe}.mid_layers.{i}.norm"
        ws += [sd[p + ".scale.weight"].reshape(512, 512), sd[p + ".shift.weight"].reshape(512, 512)]
        bs += [sd[p + ".scale.bias"], sd[p + ".shift.bias"]]
        _convnext(out, sd, f"{fe}.mid_layers.{i}", f"fe.mid{i}", True, pack_conv_split_h)       # (+ the fp16 plane of decoder precision mode 1)
    out["fe.normfilm.W"] = pack_conv_split_h(torch.cat(ws, 0).float().unsqueeze(2))
    out["fe.normfilm.b"] = _vec(torch.cat(bs, 0))
    out["osc.amps.W"] = pack_conv_split(sd["harmonic_oscillator.to_amps.weight"])
    out["osc.amps.b"] = _vec(sd["harmonic_oscillator.to_amps.bias"])
    f = "filter"
    ws, bs, post = [], [], []
    for s, c in enumerate(FILTER_CH):
        for j in range(3):
            for cc in ("c1", "c2"):
                p = f"{f}.blocks.{s}.blocks.{j}.{cc}"
                ws += [sd[p + ".to_scale.weight"].reshape(c, 512), sd[p + ".to_shift.weight"].reshape(c, 512)]
                bs += [sd[p + ".to_scale.bias"], sd[p + ".to_shift.bias"]]
                post += [torch.ones(c, device=ws[0].device), torch.zeros(c, device=ws[0].device)]
    out["flt.film.W"] = pack_conv_split(torch.cat(ws, 0).float().unsqueeze(2))
    out["flt.film.b"] = _vec(torch.cat(bs, 0))
    out["flt.film.post"] = _vec(torch.cat(post, 0))
    out["flt.in.W"] = _vec(sd[f + ".source_in.weight"])          # [8][1][7] as is: streaming kernel (csrc/filter_edge.hip)
    out["flt.in.b"] = _vec(sd[f + ".source_in.bias"])
    for i in range(4):
        # downs[0] is fused with source_in (reference layout [16][8][2]); the others run on the fp32 MFMA kernel
        out[f"flt.down{i}.W"] = _vec(sd[f"{f}.downs.{i}.weight"]) if i == 0 else pack_conv(sd[f"{f}.downs.{i}.weight"])
        out[f"flt.down{i}.b"] = _vec(sd[f"{f}.downs.{i}.bias"])
        if i >= 2:      # batch path: the strided conv as a plane GEMM (networks.hip::decoder_run)
            out[f"flt.down{i}.Wp"] = pack_conv_split_h(sd[f"{f}.downs.{i}.weight"])
    out["flt.mid.W"] = pack_conv_split_h(sd[f + ".mid_conv.conv.weight"])
    out["flt.mid.b"] = _vec(sd[f + ".mid_conv.conv.bias"])
    for i in range(4):
        pt = pack_convT_split_h if SPLIT_SCALE[i] else pack_convT          # (+ the fp16 plane of decoder precision mode 1)
        wt, bt = sd[f"{f}.ups.{i}.weight"], sd[f"{f}.ups.{i}.bias"]
        if FILTER_MODE[i] == "split":
            # ups[i] and blocks[i].input_conv (decoder.py:147,192-193) are two linear maps with nothing in between: one transposed
            # conv with the product of their weights, formed in float64 -- out[c'][t r + j] = sum_ci Wc[ci][c'][j] x[ci][t] + bc[c'],
            # Wc[ci][c'][j] = sum_c Win[c'][c] Wt[ci][c][j], bc = Win bt + bin.  The 1x1 conv over the upsampled tensor is gone.
            win = sd[f"{f}.blocks.{i}.input_conv.weight"].double()[:, :, 0]
            bc = win @ bt.double() + sd[f"{f}.blocks.{i}.input_conv.bias"].double()
            wt = torch.einsum("icj,dc->idj", wt.double(), win).float()
            bt = bc.float()
        out[f"flt.up{i}.W"], out[f"flt.up{i}.b"] = pt(wt, bt)
        if FILTER_MODE[i] == "mid":
            # the same composition for the 64-channel scale, used when its FilterBlock runs on csrc/filter_big.hip (decoder precision mode
            # 1, batch path: alive_filter_block64s_fp16 takes the residual stream like the 256-channel block); filter_mid.hip keeps the
            # input conv inside and the plain ups[i] above
            win = sd[f"{f}.blocks.{i}.input_conv.weight"].double()[:, :, 0]
            bc = win @ bt.double() + sd[f"{f}.blocks.{i}.input_conv.bias"].double()
            out[f"flt.up{i}.Wc"], out[f"flt.up{i}.bc"] = pt(torch.einsum("icj,dc->idj", wt.double(), win).float(), bc.float())
    for s in range(4):
        b = f"{f}.blocks.{s}"
        if FILTER_MODE[s] == "small":                # 16- / 8-channel scales: one fused kernel per FilterBlock
            out[f"flt.blk{s}.pack"] = pack_filter_small(sd, b)
            continue
        if FILTER_MODE[s] == "mid":                  # 64-channel scale: fused, split-bf16
            out[f"flt.blk{s}.packW"], out[f"flt.blk{s}.packB"] = pack_filter_mid(sd, b)
            for j in range(3):                       # (+ the six k5 convs as fp16 slabs of their own: csrc/filter_big.hip)
                for cc in ("c1", "c2"):
                    out[f"flt.blk{s}.{j}.{cc}.W"] = pack_conv_split_h(sd[f"{b}.blocks.{j}.{cc}.conv.conv.weight"])
                    out[f"flt.blk{s}.{j}.{cc}.b"] = _vec(sd[f"{b}.blocks.{j}.{cc}.conv.conv.bias"])
            continue
        for j in range(3):                           # (input_conv: composed into flt.up{s} above)
            for cc in ("c1", "c2"):
                out[f"flt.blk{s}.{j}.{cc}.W"] = pack_conv_split_h(sd[f"{b}.blocks.{j}.{cc}.conv.conv.weight"])
                out[f"flt.blk{s}.{j}.{cc}.b"] = _vec(sd[f"{b}.blocks.{j}.{cc}.conv.conv.bias"])
    out["flt.out.W"] = _vec(sd[f + ".source_out.weight"])        # [1][8][7] as is
    out["flt.out.b"] = _vec(sd[f + ".source_out.bias"])
    return out
