"""Drop-in for /root/reference/module/f0_estimator.py:8-34 (`estimate`: argmax class index as float Hz)."""
import torch

from . import _native as nat
from . import schema
from ._netbase import PackedNet
from ._pack import pack_f0_estimator


class F0Estimator(PackedNet):
    MODEL_ID = 1
    PREFIX = "pe."
    _pack = staticmethod(pack_f0_estimator)

    DEFAULTS = (1280, 256, 512, 4096, 4)

    def __init__(self, n_fft=1280, internal_channels=256, hidden_channels=512, output_channels=4096, num_layers=4, seed=None):
        """the reference's constructor signature (f0_estimator.py:9-14).  The fused kernels are built for its default sizes; any
        other sizes run layer by layer through the op-level entry points (module/_generic.py: same arithmetic, one launch per layer)"""
        self.sizes = (n_fft, internal_channels, hidden_channels, output_channels, num_layers)
        self.generic = self.sizes != self.DEFAULTS
        self._schema = lambda: schema.f0_estimator_schema(internal_channels, hidden_channels, output_channels, num_layers, n_fft // 2 + 1)
        super().__init__(seed)

    def _check(self, x):
        x = x.contiguous().float()
        if x.dim() != 3 or x.shape[1] != self.sizes[0] // 2 + 1:
            raise ValueError(f"F0Estimator expects [N, {self.sizes[0] // 2 + 1}, T] spectrogram bins, got {tuple(x.shape)}")
        if x.device.type != "cuda" or self._device.type != "cuda":
            raise RuntimeError("this network runs on the MI355X only: call .to('cuda') and pass CUDA tensors (no CPU path)")
        return x

    def forward(self, x):
        """x [N, 641, T] -> class logits [N, 4096, T] (f0_estimator.py:22-27).  `estimate` never stores them (the argmax sits in the
        classifier GEMM's epilogue); this is the layer-by-layer path."""
        from . import _generic
        return _generic.f0_logits(self._sd, self._check(x), self.sizes[4])

    __call__ = forward

    def estimate(self, x, downsample_factor=1, out=None):
        """x [N, 641, T] -> f0 [N, 1, T]   (out: a contiguous [N, 1, T] tensor to write into)"""
        x = self._check(x)
        n, c, t = x.shape
        if self.generic:                   # non-default sizes: logits layer by layer, then the argmax kernel (f0_estimator.py:29-34)
            from . import ops
            f0g = ops.argmax_channels(self.forward(x))
            if out is not None:
                out.copy_(f0g)
                return out
            return f0g
        L = nat.lib()
        f0 = out if out is not None else torch.empty(n, 1, t, device=x.device)
        if tuple(f0.shape) != (n, 1, t) or f0.dtype != torch.float32 or not f0.is_contiguous():
            raise ValueError("F0Estimator: out must be a contiguous fp32 [N, 1, T] tensor")
        ws = self._ws.get(L.alive_f0_estimate_workspace_bytes(n, t), x.device)
        nat.check(L.alive_f0_estimate(self.table().array, nat.ptr(x), n, t, nat.ptr(f0), nat.ptr(ws), nat.stream()),
                  "alive_f0_estimate")
        return f0
