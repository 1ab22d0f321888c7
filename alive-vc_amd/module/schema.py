"""Checkpoint schema of the three inference networks.

The state_dict key names / shapes are the drop-in boundary (ii) of DESIGN.md:
a checkpoint written by the reference's training scripts must load into this
package unchanged.  Keys and shapes follow the reference module constructors:

  content_encoder.pt  -> /root/reference/module/content_encoder.py:8-20
  f0_estimator.pt     -> /root/reference/module/f0_estimator.py:8-21
  decoder.pt          -> /root/reference/module/decoder.py:13-203
  shared blocks       -> /root/reference/module/common.py:14-92

Each entry is  name -> (shape, kind, fan_in)  where `kind` only steers the
synthetic initialiser in `synthetic.py` (no pretrained weights ship with the
reference, so all parity work runs on seeded synthetic weights).
"""
from collections import OrderedDict

N_FFT = 1280
N_BINS = N_FFT // 2 + 1          # 641
CONTENT_DIM = 768
NUM_HARMONICS = 64
SEGMENT = 320                    # samples per content frame at 16 kHz
SAMPLE_RATE = 16000
F0_CLASSES = 4096

FILTER_RATES = [2, 2, 8, 10]
FILTER_CHANNELS = [8, 16, 64, 256]


def _conv(d, name, co, ci, k, kind="conv"):
    d[name + ".weight"] = ((co, ci, k), kind, ci * k)
    d[name + ".bias"] = ((co,), "bias", ci * k)


def _convnext(d, prefix, c, h, adaptive_cond=None):
    # key order mirrors nn.Module registration order in common.py:46-52 / 66-72
    if adaptive_cond is None:
        d[prefix + ".scale"] = ((1, c, 1), "layerscale", 0)
        d[prefix + ".dw_conv.weight"] = ((c, 1, 7), "conv", 7)
        d[prefix + ".dw_conv.bias"] = ((c,), "bias", 7)
        d[prefix + ".norm.scale"] = ((1, c, 1), "gain", 0)
        d[prefix + ".norm.shift"] = ((1, c, 1), "offset", 0)
    else:
        d[prefix + ".scale"] = ((1, c, 1), "layerscale", 0)
        d[prefix + ".dw_conv.weight"] = ((c, 1, 7), "conv", 7)
        d[prefix + ".dw_conv.bias"] = ((c,), "bias", 7)
        _conv(d, prefix + ".norm.shift", c, adaptive_cond, 1)
        _conv(d, prefix + ".norm.scale", c, adaptive_cond, 1)
    _conv(d, prefix + ".pw_conv1", h, c, 1)
    _conv(d, prefix + ".pw_conv2", c, h, 1)


def content_encoder_schema(internal=512, hidden=1536, out=CONTENT_DIM, layers=4, bins=N_BINS):
    d = OrderedDict()
    _conv(d, "input_layer", internal, bins, 1)
    for i in range(layers):
        _convnext(d, f"mid_layers.{i}", internal, hidden)
    _conv(d, "output_layer", out, internal, 1)
    return d


def f0_estimator_schema(internal=256, hidden=512, out=F0_CLASSES, layers=4, bins=N_BINS):
    d = OrderedDict()
    _conv(d, "input_layer", internal, bins, 1)
    for i in range(layers):
        _convnext(d, f"mid_layers.{i}", internal, hidden)
    d["last_norm.scale"] = ((1, internal, 1), "gain", 0)
    d["last_norm.shift"] = ((1, internal, 1), "offset", 0)
    _conv(d, "output_layer", out, internal, 1)
    return d


def decoder_schema(content=CONTENT_DIM, channels=512, hidden=1536, layers=4,
                   harmonics=NUM_HARMONICS, rates=None, fchannels=None, dilations=3):
    rates = list(FILTER_RATES if rates is None else rates)
    fch = list(FILTER_CHANNELS if fchannels is None else fchannels)
    d = OrderedDict()
    fe = "feature_extractor"
    _conv(d, fe + ".input_layer", channels, content, 1)
    _conv(d, fe + ".f0_enc.c1", channels, 1, 1, kind="f0enc")
    _conv(d, fe + ".f0_enc.c2", channels, channels, 1)
    for i in range(layers):
        _convnext(d, f"{fe}.mid_layers.{i}", channels, hidden, adaptive_cond=channels)
    _conv(d, "harmonic_oscillator.to_amps", harmonics, channels, 1)
    f = "filter"
    _conv(d, f + ".source_in", fch[0], 1, 7)
    nexts = fch[1:] + [fch[-1]]
    for i, (c, cn, r) in enumerate(zip(fch, nexts, rates)):
        _conv(d, f"{f}.downs.{i}", cn, c, r)
    _conv(d, f + ".mid_conv.conv", fch[-1], fch[-1], 5)
    rch = list(reversed(fch))
    rr = list(reversed(rates))
    prevs = [rch[0]] + rch[:-1]
    for i, (c, cp, r) in enumerate(zip(rch, prevs, rr)):
        # ConvTranspose1d weight layout is [in, out, k]  (decoder.py:178)
        d[f"{f}.ups.{i}.weight"] = ((cp, c, r), "conv", cp)
        d[f"{f}.ups.{i}.bias"] = ((c,), "bias", cp)
    for i, c in enumerate(rch):
        b = f"{f}.blocks.{i}"
        _conv(d, b + ".input_conv", c, c, 1)
        for j in range(dilations):
            for cc in ("c1", "c2"):
                p = f"{b}.blocks.{j}.{cc}"
                _conv(d, p + ".conv.conv", c, c, 5)
                _conv(d, p + ".to_scale", c, channels, 1)
                _conv(d, p + ".to_shift", c, channels, 1)
    _conv(d, f + ".source_out", 1, rch[-1], 7)
    return d


def reorder_like(d, keys):
    """Return an OrderedDict with `keys` order (used to match the reference's
    registration order when writing checkpoints)."""
    return OrderedDict((k, d[k]) for k in keys)
