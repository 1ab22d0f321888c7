"""CPU ORACLE -- test infrastructure only, never the product path.

A functional PyTorch-CPU restatement of the reference's inference hot path
(uthree/ALiVE-VC), written from the operator semantics, one function per
reference function.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this file; the product package
(alive-vc_amd/) must never import it and fails loudly when its HIP library is
missing.

Pinning: the reference has no tests or golden vectors of its own, so the
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, produced by importing
/root/reference in the build container with oracle/gen_golden.py (committed)
and stored as fixtures under tests/golden/.  tests/test_oracle_golden.py
replays every fixture through this file.  The edges that live in absent
third-party packages (torchaudio resample/load/save, pyworld) are "parity
unpinned" and are not part of this oracle.

Every function works on a plain dict of tensors with the reference's
state_dict key names (prefix `p`), float32 on CPU.
"""
import math

import torch
import torch.nn.functional as F

TWO_PI = 2 * math.pi


# --------------------------------------------------------------------------
# shared blocks  (reference: module/common.py)
# --------------------------------------------------------------------------
def channel_stats_normalise(x, eps=1e-4):
    """(x - mean_C) / (std_C(unbiased) + eps)   -- common.py:21-23 / 36-38"""
    mu = x.mean(dim=1, keepdim=True)
    sigma = x.std(dim=1, keepdim=True) + eps
    return (x - mu) / sigma


def channel_norm(sd, p, x):
    """ChannelNorm.forward -- common.py:20-26"""
    return channel_stats_normalise(x) * sd[p + ".scale"] + sd[p + ".shift"]


def adaptive_channel_norm(sd, p, x, cond):
    """AdaptiveChannelNorm.forward -- common.py:35-41 (scale conv has no +1)"""
    sc = F.conv1d(cond, sd[p + ".scale.weight"], sd[p + ".scale.bias"])
    sh = F.conv1d(cond, sd[p + ".shift.weight"], sd[p + ".shift.bias"])
    return channel_stats_normalise(x) * sc + sh


def convnext1d(sd, p, x, cond=None):
    """ConvNeXt1d.forward (common.py:54-62) / AdaptiveConvNeXt1d.forward (74-82)"""
    c = x.shape[1]
    y = F.conv1d(x, sd[p + ".dw_conv.weight"], sd[p + ".dw_conv.bias"], padding=3, groups=c)
    if cond is None:
        y = channel_norm(sd, p + ".norm", y)
    else:
        y = adaptive_channel_norm(sd, p + ".norm", y, cond)
    y = F.conv1d(y, sd[p + ".pw_conv1.weight"], sd[p + ".pw_conv1.bias"])
    y = F.gelu(y)
    y = F.conv1d(y, sd[p + ".pw_conv2.weight"], sd[p + ".pw_conv2.bias"])
    return y * sd[p + ".scale"] + x


def causal_conv1d(sd, p, x, dilation=1):
    """CausalConv1d.forward -- common.py:85-92: reflect-pad (k-1)*d on the left only."""
    w = sd[p + ".conv.weight"]
    k = w.shape[2]
    x = F.pad(x, (k * dilation - dilation, 0), mode="reflect")
    return F.conv1d(x, w, sd[p + ".conv.bias"], dilation=dilation)


def match_features(source, reference, k=4, alpha=0.0, return_indices=False):
    """match_features -- common.py:96-109 (cosine kNN regression)."""
    s = source.transpose(1, 2)
    r = reference.transpose(1, 2)
    sn = torch.norm(s, dim=2, keepdim=True)
    rn = torch.norm(r, dim=2, keepdim=True)
    cos = torch.bmm(s / sn, (r / rn).transpose(1, 2))
    best = torch.topk(cos, k, dim=2)
    picked = torch.stack([r[n][best.indices[n]] for n in range(s.shape[0])], dim=0)
    out = picked.mean(dim=2).transpose(1, 2)
    out = out * (1 - alpha) + source * alpha
    if return_indices:
        return out, best.indices, cos
    return out


def voice_library_match(tokens, source, k=4, alpha=0.0):
    """VoiceLibrary.match -- voice_library.py:15-33 (tokens expanded over batch)."""
    ref = tokens.expand(source.shape[0], tokens.shape[1], tokens.shape[2])
    return match_features(source, ref, k, alpha)


# --------------------------------------------------------------------------
# front end  (module/spectrogram.py, content_encoder.py, f0_estimator.py)
# --------------------------------------------------------------------------
def spectrogram(x):
    """spectrogram -- spectrogram.py:5-10: rect-window |STFT| 1280/320, drop last frame."""
    s = torch.stft(x.to(torch.float), 1280, 320, 1280, center=True, return_complex=True).abs()
    return s.to(x.dtype)[:, :, :-1]


def content_encoder(sd, spec, p=""):
    """ContentEncoder.forward -- content_encoder.py:21-25"""
    x = F.conv1d(spec, sd[p + "input_layer.weight"], sd[p + "input_layer.bias"])
    i = 0
    while f"{p}mid_layers.{i}.scale" in sd:
        x = convnext1d(sd, f"{p}mid_layers.{i}", x)
        i += 1
    return F.conv1d(x, sd[p + "output_layer.weight"], sd[p + "output_layer.bias"])


def f0_logits(sd, spec, p=""):
    """F0Estimator.forward -- f0_estimator.py:22-27"""
    x = F.conv1d(spec, sd[p + "input_layer.weight"], sd[p + "input_layer.bias"])
    i = 0
    while f"{p}mid_layers.{i}.scale" in sd:
        x = convnext1d(sd, f"{p}mid_layers.{i}", x)
        i += 1
    x = channel_norm(sd, p + "last_norm", x)
    return F.conv1d(x, sd[p + "output_layer.weight"], sd[p + "output_layer.bias"])


def f0_estimate(sd, spec, p=""):
    """F0Estimator.estimate -- f0_estimator.py:29-34: argmax class index as float Hz."""
    lg = f0_logits(sd, spec, p)
    return torch.argmax(lg, dim=1).to(lg.dtype).unsqueeze(1)


# --------------------------------------------------------------------------
# decoder  (module/decoder.py)
# --------------------------------------------------------------------------
def f0_encoder(sd, p, f0):
    """F0Encoder.forward -- decoder.py:20-24"""
    x = F.conv1d(f0, sd[p + ".c1.weight"], sd[p + ".c1.bias"])
    x = torch.sin(x)
    return F.conv1d(x, sd[p + ".c2.weight"], sd[p + ".c2.bias"])


def feature_extractor(sd, p, x, f0):
    """FeatureExtractor.forward -- decoder.py:43-48"""
    x = F.conv1d(x, sd[p + ".input_layer.weight"], sd[p + ".input_layer.bias"])
    c = f0_encoder(sd, p + ".f0_enc", f0)
    i = 0
    while f"{p}.mid_layers.{i}.scale" in sd:
        x = convnext1d(sd, f"{p}.mid_layers.{i}", x, cond=c)
        i += 1
    return x


def harmonic_oscillator(sd, p, x, f0, phi=0, crop0=0, segment=320, sample_rate=16000,
                        return_debug=False):
    """HarmonicOscillator.forward -- decoder.py:66-102.
    exp-amplitudes, 64 integer multiples of f0, x320 linear upsampling,
    cumulative phase (CPU cumsum: fp64 accumulate, each prefix rounded to
    fp32), phase origin at column crop0, sin, asin carry, mean over harmonics."""
    w = sd[p + ".to_amps.weight"]
    nh = w.shape[0]
    n, _, lf = x.shape
    lw = lf * segment
    amps = torch.exp(F.conv1d(x, w, sd[p + ".to_amps.bias"]))
    mul = (torch.arange(nh) + 1).view(1, nh, 1).expand(n, nh, lf)
    formants = f0 * mul
    formants = F.interpolate(formants, lw, mode="linear")
    amps = F.interpolate(amps, lw, mode="linear")
    dt = torch.cumsum(formants / sample_rate, dim=2)
    dt = dt - dt[:, :, crop0].unsqueeze(2)
    theta = TWO_PI * dt + phi
    harmonics = torch.sin(theta)
    phi_out = torch.asin(harmonics)
    wave = (harmonics * amps).mean(dim=1, keepdim=True)
    if return_debug:
        return wave, phi_out, dict(dt=dt, theta=theta, amps=amps, formants=formants)
    return wave, phi_out


def _film(sd, p, x, c):
    """scale/shift of ModulatedCausalConv1d -- decoder.py:112-117"""
    scale = F.conv1d(c, sd[p + ".to_scale.weight"], sd[p + ".to_scale.bias"]) + 1
    shift = F.conv1d(c, sd[p + ".to_shift.weight"], sd[p + ".to_shift.bias"])
    scale = F.interpolate(scale, x.shape[2], mode="linear")
    shift = F.interpolate(shift, x.shape[2], mode="linear")
    return x * scale + shift


def modulated_causal_conv(sd, p, x, c, dilation):
    """ModulatedCausalConv1d.forward -- decoder.py:112-119"""
    return causal_conv1d(sd, p + ".conv", _film(sd, p, x, c), dilation)


def filter_res_block(sd, p, x, c, dilation):
    """FilterResBlock.forward -- decoder.py:128-134"""
    y = modulated_causal_conv(sd, p + ".c1", F.gelu(x), c, dilation)
    y = modulated_causal_conv(sd, p + ".c2", F.gelu(y), c, dilation)
    return y + x


def filter_block(sd, p, x, c):
    """FilterBlock.forward -- decoder.py:146-150 (dilations 1, 2, 4, ...)"""
    x = F.conv1d(x, sd[p + ".input_conv.weight"], sd[p + ".input_conv.bias"])
    j = 0
    while f"{p}.blocks.{j}.c1.conv.conv.weight" in sd:
        x = filter_res_block(sd, f"{p}.blocks.{j}", x, c, 2 ** j)
        j += 1
    return x


def source_filter(sd, p, src, c):
    """Filter.forward -- decoder.py:184-195 (U-Net over the oscillator waveform)."""
    x = F.conv1d(src, sd[p + ".source_in.weight"], sd[p + ".source_in.bias"], padding=3)
    skips = []
    i = 0
    while f"{p}.downs.{i}.weight" in sd:
        w = sd[f"{p}.downs.{i}.weight"]
        x = F.conv1d(x, w, sd[f"{p}.downs.{i}.bias"], stride=w.shape[2])
        skips.append(x)
        i += 1
    x = causal_conv1d(sd, p + ".mid_conv", x)
    for i, s in enumerate(reversed(skips)):
        w = sd[f"{p}.ups.{i}.weight"]
        x = F.conv_transpose1d(x + s, w, sd[f"{p}.ups.{i}.bias"], stride=w.shape[2])
        x = filter_block(sd, f"{p}.blocks.{i}", x, c)
    return F.conv1d(x, sd[p + ".source_out.weight"], sd[p + ".source_out.bias"], padding=3)


def decoder(sd, x, f0, phi=0, crop0=0, p=""):
    """Decoder.forward -- decoder.py:205-210 at harmonics_scale == 1 (the only
    value for which the reference's tuple*scale expression runs)."""
    feats = feature_extractor(sd, p + "feature_extractor", x, f0)
    src, phi_out = harmonic_oscillator(sd, p + "harmonic_oscillator", feats, f0, phi, crop0)
    out = source_filter(sd, p + "filter", src, feats).squeeze(1)
    return out, phi_out


# --------------------------------------------------------------------------
# driver loop bodies  (inference.py / realtime_inference.py)
# --------------------------------------------------------------------------
def pitch_transform_offline(f0, pitch_shift=0.0, intonation=1.0, f0_rate=1.0):
    """inference.py:119-126,130 -- per-window mean pitch over finite entries."""
    pitch = 12 * torch.log2(f0 / 440) - 9
    finite = torch.logical_not(torch.logical_or(pitch.isinf(), pitch.isnan()))
    mean_pitch = pitch.masked_select(finite).mean()
    pitch = mean_pitch + (pitch - mean_pitch) * intonation + pitch_shift
    f0 = 440 * 2 ** ((pitch + 9) / 12)
    f0[torch.logical_or(f0.isnan(), f0.isinf())] = 0
    return f0 * f0_rate


def pitch_transform_realtime(f0, pitch_shift=0.0):
    """realtime_inference.py:156-163 (f0_rate is applied by the caller before)."""
    pitch = 12 * torch.log2(f0 / 440) - 9
    pitch = pitch + pitch_shift
    f0 = 440 * 2 ** ((pitch + 9) / 12)
    f0[torch.logical_or(f0.isnan(), f0.isinf())] = 0
    return f0


def make_windows(wf, chunk=48000):
    """inference.py:94-101 -- windows of 3*chunk at hop chunk over the padded signal."""
    total = wf.shape[1]
    wf = torch.cat([wf, torch.zeros(1, chunk * 3)], dim=1)
    wf = F.pad(wf.unsqueeze(1).unsqueeze(1), (chunk, chunk, 0, 0))
    win = F.unfold(wf, (1, chunk * 3), stride=chunk)       # [1, 3*chunk, n_windows]
    return win.transpose(1, 2)[0], total                   # [n_windows, 3*chunk]


def convert_window(ce, pe, dec, window, tgt, k=4, alpha=0.0, pitch_shift=0.0,
                   intonation=1.0, f0_rate=1.0):
    """one iteration of inference.py:106-130; `window` is [1, 3*chunk]."""
    spec = spectrogram(window)
    f0 = f0_estimate(pe, spec)
    f0 = pitch_transform_offline(f0, pitch_shift, intonation, f0_rate)
    feat = content_encoder(ce, spec)
    feat = match_features(feat, tgt, k=k, alpha=alpha)
    wav, _ = decoder(dec, feat, f0)
    return wav


def convert_utterance(ce, pe, dec, wf, tgt, chunk=48000, **kw):
    """inference.py:94-135 without file I/O / resampling (those live in absent
    torchaudio: parity unpinned)."""
    windows, total = make_windows(wf, chunk)
    out = []
    for w in windows:
        wav = convert_window(ce, pe, dec, w.unsqueeze(0), tgt, **kw)
        out.append(wav[:, chunk:-chunk])
    return torch.cat(out, dim=1)[:, :total]


def realtime_geometry(chunk, buffersize, output_sr=16000):
    """realtime_inference.py:122-126"""
    internal_chunk = int(chunk * (16000 / output_sr))
    center = int(internal_chunk * buffersize) // 2
    return center - internal_chunk // 2, center + internal_chunk // 2


def realtime_step(ce, pe, dec, ring, tgt, phi, begin, end, k=4, alpha=0.0,
                  pitch_shift=0.0, f0_rate=1.0):
    """realtime_inference.py:146-167 at isr == osr == 16000, gain 0 dB.
    ring: float32 [1, buffersize*chunk]; returns (wave[1,L], phi_next[1,64,1])."""
    spec = spectrogram(ring)
    content = content_encoder(ce, spec)
    f0 = f0_estimate(pe, spec) * f0_rate
    f0 = pitch_transform_realtime(f0, pitch_shift)
    content = match_features(content, tgt, k=k, alpha=alpha)
    wave, phi_out = decoder(dec, content, f0, phi=phi, crop0=begin)
    return wave, phi_out[:, :, end].unsqueeze(2)


# ---- torchaudio.functional.resample / gain (Appendix C of SURVEY.md: restated from the public algorithm, torchaudio is
# not in the reference tree -> "parity unpinned"; used to check csrc/audio.hip) ---------------------------------------
def resample_filter(orig, new, width_param=6, rolloff=0.99):
    """_get_sinc_resample_kernel(sinc_interp_hann): -> (kernel float32 [new, 1, 2*width + orig], width)"""
    import math
    base = min(orig, new) * rolloff
    width = math.ceil(width_param * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp(-width_param, width_param)
    window = torch.cos(t * math.pi / width_param / 2) ** 2
    t = t * math.pi
    scale = base / orig
    k = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * scale
    return k.float(), width


def resample(waveform, orig_freq, new_freq):
    """_apply_sinc_resample_kernel: pad, strided conv with the filter bank, interleave the phases, trim"""
    import math
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    kernel, width = resample_filter(orig, new)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1])
    length = x.shape[1]
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kernel, stride=orig)            # [B, new, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = math.ceil(new * length / orig)
    return y[..., :target].reshape(shape[:-1] + (target,))


def gain(waveform, gain_db=1.0):
    return waveform if gain_db == 0 else waveform * (10 ** (gain_db / 20))
