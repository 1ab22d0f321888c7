"""WAV I/O, resampling and gain for the CLI edges of the path.

The reference uses torchaudio for these (inference.py:88-91,136-142; realtime_inference.py:146-147,
173-175); torchaudio is not part of this image and none of its arithmetic is pinned by reference
tests, so csrc/audio.hip restates the PUBLIC algorithm of torchaudio.functional.resample
(sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99; checked against the torch formulation kept in
oracle/alive_oracle.py) and this file documents its own file format choice: "parity unpinned" (DESIGN.md).  Reads PCM 8/16/24/32 and float32 WAV; writes float32 WAV by
default (lossless w.r.t. the 1e-3 RMS comparison) or PCM16.
"""
import math
import struct

import numpy as np
import torch


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


_filters = {}


def _reduced(orig_freq, new_freq):
    g = math.gcd(int(orig_freq), int(new_freq))
    return int(orig_freq) // g, int(new_freq) // g


def resample(waveform, orig_freq, new_freq, pre_gain_db=0.0, post_gain_db=0.0):
    """torchaudio.functional.resample on the device (csrc/audio.hip: polyphase windowed sinc), waveform [..., time];
    identity when the rates match.  The optional gains are torchaudio.functional.gain applied to the input / output
    inside the same kernel (bitwise the same as separate multiplies)."""
    from . import _native as nat
    orig, new = _reduced(orig_freq, new_freq)
    pre = float(10 ** (pre_gain_db / 20)) if pre_gain_db != 0 else 1.0
    post = float(10 ** (post_gain_db / 20)) if post_gain_db != 0 else 1.0
    if orig == new:
        if pre * post == 1.0:
            return waveform
        return gain(gain(waveform, pre_gain_db), post_gain_db)
    L_ = nat.lib()
    key = (orig, new, str(waveform.device))
    if key not in _filters:
        f = torch.empty(new * L_.alive_resample_taps(orig, new), device=waveform.device)
        nat.check(L_.alive_resample_filter(orig, new, nat.ptr(f), nat.stream()), "alive_resample_filter")
        torch.cuda.current_stream(waveform.device).synchronize()    # once per rate pair: other streams will read it
        _filters[key] = f
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1]).contiguous().float()
    lout = int(L_.alive_resample_length(x.shape[1], orig, new))
    y = torch.empty(x.shape[0], lout, device=x.device)
    nat.check(L_.alive_resample(nat.ptr(x), x.shape[0], x.shape[1], orig, new, nat.ptr(_filters[key]), pre, post,
                                nat.ptr(y), lout, nat.stream()), "alive_resample")
    return y.reshape(shape[:-1] + (lout,))


def gain(waveform, gain_db=1.0):
    """torchaudio.functional.gain: multiply by 10^(dB/20)"""
    if gain_db == 0:
        return waveform
    return waveform * (10 ** (gain_db / 20))


def pcm16_to_float(data_i16):
    """int16 device tensor -> float32 / 32768 (realtime_inference.py:139-140)"""
    from . import _native as nat
    out = torch.empty(data_i16.shape, dtype=torch.float32, device=data_i16.device)
    nat.check(nat.lib().alive_pcm16_to_float(nat.ptr(data_i16), data_i16.numel(), nat.ptr(out), nat.stream()), "alive_pcm16_to_float")
    return out


def float_to_pcm16(wave):
    """float32 device tensor -> int16 by the C cast of numpy's astype: truncate toward zero, no clipping
    (realtime_inference.py:180-183)"""
    from . import _native as nat
    wave = wave.contiguous()
    out = torch.empty(wave.shape, dtype=torch.int16, device=wave.device)
    nat.check(nat.lib().alive_float_to_pcm16(nat.ptr(wave), wave.numel(), nat.ptr(out), nat.stream()), "alive_float_to_pcm16")
    return out
