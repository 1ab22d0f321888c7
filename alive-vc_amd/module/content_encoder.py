"""Drop-in for /root/reference/module/content_encoder.py:8-25 (default sizes: 641 -> 512 -> 4x ConvNeXt1d(512,1536) -> 768)."""
import torch

from . import _native as nat
from . import schema
from ._netbase import PackedNet
from ._pack import pack_content_encoder


class ContentEncoder(PackedNet):
    MODEL_ID = 0
    PREFIX = "ce."
    _pack = staticmethod(pack_content_encoder)

    DEFAULTS = (1280, 512, 1536, 768, 4)

    def __init__(self, n_fft=1280, internal_channels=512, hidden_channels=1536, output_channels=768, num_layers=4, seed=None):
        """the reference's constructor signature (content_encoder.py:9-14).  The fused kernels are built for its default sizes -- the
        only ones inference.py / realtime_inference.py / generate_voice_library.py ever construct; any other sizes run layer by
        layer through the op-level entry points (module/_generic.py: same arithmetic, one launch per layer)"""
        self.sizes = (n_fft, internal_channels, hidden_channels, output_channels, num_layers)
        self.generic = self.sizes != self.DEFAULTS
        self._schema = lambda: schema.content_encoder_schema(internal_channels, hidden_channels, output_channels, num_layers, n_fft // 2 + 1)
        super().__init__(seed)

    def forward(self, x, out=None):
        """x [N, 641, T] -> [N, 768, T]   (out: a contiguous [N, 768, T] tensor to write into, e.g. a batch slice)"""
        x = x.contiguous().float()
        n, c, t = x.shape
        if c != self.sizes[0] // 2 + 1:
            raise ValueError(f"ContentEncoder expects {self.sizes[0] // 2 + 1} spectrogram bins, got {c}")
        if self.generic:
            from . import _generic
            if x.device.type != "cuda" or self._device.type != "cuda":
                raise RuntimeError("this network runs on the MI355X only: call .to('cuda') and pass CUDA tensors (no CPU path)")
            y = _generic.content_encoder(self._sd, x, self.sizes[4])
            if out is not None:
                out.copy_(y)
                return out
            return y
        L = nat.lib()
        if out is None:
            out = torch.empty(n, schema.CONTENT_DIM, t, device=x.device)
        elif tuple(out.shape) != (n, schema.CONTENT_DIM, t) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("ContentEncoder: out must be a contiguous fp32 [N, 768, T] tensor")
        ws = self._ws.get(L.alive_content_encoder_workspace_bytes(n, t), x.device)
        nat.check(L.alive_content_encoder(self.table().array, nat.ptr(x), n, t, nat.ptr(out), nat.ptr(ws), nat.stream()),
                  "alive_content_encoder")
        return out

    __call__ = forward
