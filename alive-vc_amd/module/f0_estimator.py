"""Drop-in for /root/reference/module/f0_estimator.py:8-34 (`estimate`: argmax class index as float Hz)."""
import torch

from . import _native as nat
from . import schema
from ._netbase import PackedNet
from ._pack import pack_f0_estimator


class F0Estimator(PackedNet):
    MODEL_ID = 1
    PREFIX = "pe."
    _schema = staticmethod(schema.f0_estimator_schema)
    _pack = staticmethod(pack_f0_estimator)

    def __init__(self, n_fft=1280, internal_channels=256, hidden_channels=512, output_channels=4096, num_layers=4, seed=None):
        """the reference's constructor signature (f0_estimator.py:9-14); the kernels are built for its default sizes"""
        if (n_fft, internal_channels, hidden_channels, output_channels, num_layers) != (1280, 256, 512, 4096, 4):
            raise ValueError("F0Estimator: this build implements the reference's default architecture only "
                             "(n_fft 1280, 256 / 512 channels, 4096 classes, 4 layers)")
        super().__init__(seed)

    def estimate(self, x, downsample_factor=1, out=None):
        """x [N, 641, T] -> f0 [N, 1, T]   (out: a contiguous [N, 1, T] tensor to write into)"""
        x = x.contiguous().float()
        n, c, t = x.shape
        if c != schema.N_BINS:
            raise ValueError(f"F0Estimator expects 641 spectrogram bins, got {c}")
        L = nat.lib()
        f0 = out if out is not None else torch.empty(n, 1, t, device=x.device)
        if tuple(f0.shape) != (n, 1, t) or f0.dtype != torch.float32 or not f0.is_contiguous():
            raise ValueError("F0Estimator: out must be a contiguous fp32 [N, 1, T] tensor")
        ws = self._ws.get(L.alive_f0_estimate_workspace_bytes(n, t), x.device)
        nat.check(L.alive_f0_estimate(self.table().array, nat.ptr(x), n, t, nat.ptr(f0), nat.ptr(ws), nat.stream()),
                  "alive_f0_estimate")
        return f0
