"""Drop-in for /root/reference/module/spectrogram.py:5-10 -- rect-window magnitude STFT
(n_fft 1280, hop 320, centre reflect pad, last frame dropped) as a DFT GEMM on the f32 MFMA."""
import torch

from . import _native as nat

_basis = {}
_ws = nat.Workspace()


def dft_basis(dev):
    """the device's DFT basis buffer (fp32 image + two plane-packed bf16 images), built once per device"""
    L = nat.lib()
    b = _basis.get(str(dev))
    if b is None:
        b = torch.empty(L.alive_dft_basis_bytes() // 4, dtype=torch.float32, device=dev)
        nat.check(L.alive_dft_basis(nat.ptr(b), nat.stream()), "alive_dft_basis")
        torch.cuda.current_stream(dev).synchronize()      # once per device: other streams will read it (module/pipeline.py)
        _basis[str(dev)] = b
    return b


def spectrogram(x):
    """x [N, L] -> [N, 641, L // 320]"""
    dtype = x.dtype
    x = x.contiguous().float()
    n, l = x.shape
    if l <= 640:
        raise ValueError(f"spectrogram needs more than 640 samples (reflect padding), got {l}")
    L = nat.lib()
    dev = x.device
    b = dft_basis(dev)
    out = torch.empty(n, 641, l // 320, dtype=torch.float32, device=dev)
    ws = _ws.get(L.alive_spectrogram_workspace_bytes(n, l), dev)
    nat.check(L.alive_spectrogram(nat.ptr(b), nat.ptr(x), n, l, nat.ptr(out), nat.ptr(ws), nat.stream()), "alive_spectrogram")
    return out.to(dtype)
