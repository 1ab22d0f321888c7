"""WAV I/O, resampling and gain for the CLI edges of the path.

The reference uses torchaudio for these (inference.py:88-91,136-142; realtime_inference.py:146-147,
173-175); torchaudio is not part of this image and none of its arithmetic is pinned by reference
tests, so this file restates the PUBLIC algorithm of torchaudio.functional.resample
(sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99) and documents its own file format choice:
"parity unpinned" (DESIGN.md).  Reads PCM 8/16/24/32 and float32 WAV; writes float32 WAV by
default (lossless w.r.t. the 1e-3 RMS comparison) or PCM16.
"""
import math
import struct

import numpy as np
import torch
import torch.nn.functional as F


def load(path):
    """-> (float32 tensor [channels, samples] in [-1, 1], sample_rate)"""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:            # WAVE_FORMAT_EXTENSIBLE: sub-format GUID
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, _, _, bits = fmt
    if tag == 3 and bits == 32:
        x = np.frombuffer(pcm, dtype="<f4").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm, dtype="<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = v.astype(np.float32) / 8388608.0
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported WAV format tag {tag} / {bits} bits")
    x = x[: len(x) // ch * ch].reshape(-1, ch).T
    return torch.from_numpy(np.ascontiguousarray(x)), sr


def save(path, src, sample_rate, encoding="float32"):
    """src float32 [channels, samples]"""
    x = src.detach().cpu().float().numpy().T
    ch = x.shape[1]
    if encoding == "float32":
        tag, bits, body = 3, 32, np.ascontiguousarray(x, dtype="<f4").tobytes()
    elif encoding == "pcm16":
        tag, bits = 1, 16
        body = np.clip(np.round(x * 32768.0), -32768, 32767).astype("<i2").tobytes()
    else:
        raise ValueError(encoding)
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(body), b"WAVE", b"fmt ", 16, tag, ch, sample_rate,
                      sample_rate * ch * bits // 8, ch * bits // 8, bits, b"data", len(body))
    with open(path, "wb") as f:
        f.write(hdr + body)


_kernels = {}


def _sinc_kernel(orig, new, device, width_param=6, rolloff=0.99):
    key = (orig, new, str(device))
    if key not in _kernels:
        base = min(orig, new) * rolloff
        width = math.ceil(width_param * orig / base)
        idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
        t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
        t = (t * base).clamp(-width_param, width_param)
        window = torch.cos(t * math.pi / width_param / 2) ** 2
        t = t * math.pi
        scale = base / orig
        k = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * scale
        _kernels[key] = (k.float().to(device), width)
    return _kernels[key]


def resample(waveform, orig_freq, new_freq):
    """polyphase windowed-sinc resampling, waveform [..., time]; identity when the rates match."""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    kernel, width = _sinc_kernel(orig, new, waveform.device)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1])
    length = x.shape[1]
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kernel, stride=orig)            # [B, new, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = math.ceil(new * length / orig)
    return y[..., :target].reshape(shape[:-1] + (target,))


def gain(waveform, gain_db=1.0):
    """torchaudio.functional.gain: multiply by 10^(dB/20)"""
    if gain_db == 0:
        return waveform
    return waveform * (10 ** (gain_db / 20))
