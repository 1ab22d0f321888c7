"""Bit-reproducible synthetic weights and inputs.

The reference ships no pretrained checkpoints (its .gitignore excludes *.pt),
so bench / smoke / parity all run on seeded synthetic weights.  The generator
is a counter-based integer hash (splitmix64) followed by IEEE basic operations
only, so the same (name, seed) produces the same bits on every machine and
numpy version -- `torch.randn` streams are not a cross-version contract.
Committed SHA-256 digests in tests/golden/weights_sha256.json pin it.

Initialisation scales follow what a freshly constructed reference module would
have (PyTorch Conv1d default: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and
bias; F0Encoder.c1.weight ~ N(0, 0.3): /root/reference/module/decoder.py:18;
ChannelNorm scale 1 / shift 0: common.py:17-18; ConvNeXt layer-scale 1/4:
content_encoder.py:17), with a small deterministic jitter on the affine
parameters so that parity tests exercise them.
"""
import hashlib
from collections import OrderedDict

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _name_seed(name: str, seed: int) -> int:
    h = 0xCBF29CE484222325
    for b in (name + "#" + str(seed)).encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(name: str, seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n doubles in [0,1): top 32 bits of splitmix64(counter) * 2^-32 (exact)."""
    base = np.uint64(_name_seed(name, seed))
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)
               + base + np.uint64(stream) * np.uint64(0xA24BAED4963EE407))
    z = _splitmix64(ctr)
    return (z >> np.uint64(32)).astype(np.float64) * (1.0 / 4294967296.0)


def normalish(name: str, seed: int, n: int) -> np.ndarray:
    """Irwin-Hall(12) - 6: variance-1, exact dyadic arithmetic, no libm."""
    acc = np.zeros(n, dtype=np.float64)
    for s in range(12):
        acc += uniform01(name, seed, n, stream=s + 1)
    return acc - 6.0


def make_tensor(name, shape, kind, fan_in, seed) -> torch.Tensor:
    n = int(np.prod(shape))
    if kind in ("conv", "bias"):
        b = 1.0 / np.sqrt(float(max(fan_in, 1)))
        v = (uniform01(name, seed, n) - 0.5) * (2.0 * b)
    elif kind == "f0enc":
        v = normalish(name, seed, n) * 0.3
    elif kind == "layerscale":
        v = 0.25 + (uniform01(name, seed, n) - 0.5) * 0.05
    elif kind == "gain":
        v = 1.0 + (uniform01(name, seed, n) - 0.5) * 0.2
    elif kind == "offset":
        v = (uniform01(name, seed, n) - 0.5) * 0.2
    else:
        raise ValueError(kind)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def make_state_dict(schema, seed: int, prefix: str = ""):
    sd = OrderedDict()
    for name, (shape, kind, fan_in) in schema.items():
        sd[name] = make_tensor(prefix + name, shape, kind, fan_in, seed)
    return sd


def state_dict_digest(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def gaussian(name: str, seed: int, shape, scale: float = 1.0) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy((normalish(name, seed, n) * scale).astype(np.float32).reshape(shape))


def make_library(M: int, seed: int, dim: int = 768, chunk: int = 65536) -> torch.Tensor:
    """Synthetic voice library tokens[1, dim, M] (format of
    /root/reference/generate_voice_library.py:42).  Generated column-block-wise
    so that a 1 M-vector bank never needs more than one chunk of fp64 scratch;
    column m depends only on (seed, m), so any shard can be built locally."""
    out = torch.empty(1, dim, M, dtype=torch.float32)
    for m0 in range(0, M, chunk):
        m1 = min(M, m0 + chunk)
        blk = normalish(f"library.{m0 // chunk}", seed, dim * (m1 - m0))
        out[0, :, m0:m1] = torch.from_numpy(blk.astype(np.float32).reshape(dim, m1 - m0))
    return out


def make_waveform(n_samples: int, seed: int, name: str = "wave") -> torch.Tensor:
    """0.1 * N(0,1) noise plus a few partials, mono [1, n]."""
    x = normalish(name, seed, n_samples) * 0.1
    t = np.arange(n_samples, dtype=np.float64)
    for f in (110.0, 220.0, 331.0):
        # sin via numpy libm is not bit-portable, so quantise the partial to 2^-20
        # steps: parity inputs are always compared as data, never re-derived.
        x += 0.2 * np.round(np.sin(2 * np.pi * f * t / 16000.0) * 1048576.0) / 1048576.0
    return torch.from_numpy(x.astype(np.float32).reshape(1, n_samples))


def make_voiced(n_samples: int, seed: int, f_mid: float = 140.0):
    """A voiced-speech-like test signal (round 6: parity fixtures beyond `make_waveform`'s noise + three partials): a stack of 24
    harmonics (amplitude 1/h, a fixed phase per harmonic) on a gliding fundamental f_mid +- 30 % with a syllable-rate amplitude
    envelope, plus 0.01 * N(0,1) breath noise.  Returns (wave [1, n] float32, f0 contour [n] float64 in Hz).  Like `make_waveform`
    the sines are quantised to 2^-20 (numpy's libm is not a bit-level contract); fixtures store the signal as data."""
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    f0 = f_mid * (1.0 + 0.3 * np.sin(2 * np.pi * 0.7 * t + 0.1 * seed))
    ph = 2 * np.pi * np.cumsum(f0) / 16000.0
    env = 0.15 + 0.85 * (0.5 * (1.0 + np.sin(2 * np.pi * 2.3 * t))) ** 2
    x = np.zeros(n_samples, dtype=np.float64)
    for h in range(1, 25):
        x += (0.5 / h) * np.round(np.sin(h * ph + 0.37 * h) * 1048576.0) / 1048576.0
    x = 0.25 * env * x + 0.01 * normalish("voiced.noise", seed, n_samples)
    return torch.from_numpy(x.astype(np.float32).reshape(1, n_samples)), f0


SCALED_KEYS = ("to_scale.weight", "to_shift.weight", "pw_conv1.weight", "pw_conv2.weight")


def scale_weights(sd, factor: float, keys=SCALED_KEYS):
    """A copy of `sd` with every FiLM projection (decoder.py:109-110) and ConvNeXt pointwise weight (common.py:50-51, 70-71) multiplied
    by `factor`: the Filter has no normalisation layer and its FiLM gains multiply activations, so this is the cheap stand-in for a
    trained checkpoint whose activations are an order of magnitude above a fresh initialisation's (round 6 fixtures `full_T450_x4`)."""
    out = OrderedDict()
    for k, v in sd.items():
        out[k] = v * factor if k.endswith(keys) else v.clone()
    return out


# weight sets of the round-6 reference fixtures tests/golden/full_T450_<tag>.npz: tag -> (weight seed, factor of scale_weights, input seed)
FIXTURE_SETS = {"s3": (3, 1.0, 7), "s5": (5, 1.0, 8), "x4": (2, 4.0, 9)}


def fixture_state_dicts(tag: str, schema=None):
    """(content_encoder, f0_estimator, decoder) state_dicts of fixture set `tag` (FIXTURE_SETS), as oracle/gen_golden.py loaded them
    into the reference."""
    if schema is None:
        from . import schema
    seed, fac, _ = FIXTURE_SETS[tag]
    sds = [make_state_dict(s, seed, p) for s, p in ((schema.content_encoder_schema(), "ce."), (schema.f0_estimator_schema(), "pe."),
                                                    (schema.decoder_schema(), "dec."))]
    if fac != 1.0:
        sds = [scale_weights(sd, fac) for sd in sds]
    return sds
