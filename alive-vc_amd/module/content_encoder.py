"""Drop-in for /root/reference/module/content_encoder.py:8-25 (default sizes: 641 -> 512 -> 4x ConvNeXt1d(512,1536) -> 768)."""
import torch

from . import _native as nat
from . import schema
from ._netbase import PackedNet
from ._pack import pack_content_encoder


class ContentEncoder(PackedNet):
    MODEL_ID = 0
    PREFIX = "ce."
    _schema = staticmethod(schema.content_encoder_schema)
    _pack = staticmethod(pack_content_encoder)

    def forward(self, x):
        """x [N, 641, T] -> [N, 768, T]"""
        x = x.contiguous().float()
        n, c, t = x.shape
        if c != schema.N_BINS:
            raise ValueError(f"ContentEncoder expects 641 spectrogram bins, got {c}")
        L = nat.lib()
        out = torch.empty(n, schema.CONTENT_DIM, t, device=x.device)
        ws = self._ws.get(L.alive_content_encoder_workspace_bytes(n, t), x.device)
        nat.check(L.alive_content_encoder(self.table().array, nat.ptr(x), n, t, nat.ptr(out), nat.ptr(ws), nat.stream()),
                  "alive_content_encoder")
        return out

    __call__ = forward
