"""Drop-in for /root/reference/module/decoder.py:198-210 (FeatureExtractor -> HarmonicOscillator -> Filter).

forward(x, f0, phi=0, harmonics_scale=1, crop=(0,-1)) -> (wave[N, 320*Lf], phi)

The reference returns phi[N,64,Lw] = asin(sin(theta)) for every sample although its only
caller reads one column (realtime_inference.py:167).  Here `phi` is a lazy column view:
indexing `phi[:, :, c]` for the column requested through `phi_col` returns [N,64]; by
default phi_col = crop[1] (the realtime caller's `end_of_output`), or the last sample.
"""
import torch

from . import _native as nat
from . import schema
from ._netbase import PackedNet
from ._pack import pack_decoder


class PhaseColumn:
    """Stands in for the reference's full phi tensor: holds asin(sin(theta)) at one column."""

    def __init__(self, col, values, lw):
        self.col, self.values, self.lw = col, values, lw

    def __getitem__(self, key):
        if isinstance(key, tuple) and len(key) == 3 and key[0] == slice(None) and key[1] == slice(None):
            c = key[2] if key[2] >= 0 else self.lw + key[2]
            if c == self.col:
                return self.values
        raise IndexError(f"only column {self.col} of phi was materialised (pass phi_col=... to Decoder.forward)")


class Decoder(PackedNet):
    MODEL_ID = 2
    PREFIX = "dec."
    _schema = staticmethod(schema.decoder_schema)
    _pack = staticmethod(pack_decoder)

    def forward(self, x, f0, phi=0, harmonics_scale=1, crop=(0, -1), phi_col=None, out=None):
        if harmonics_scale != 1:
            # the reference multiplies the oscillator's return TUPLE by this value (decoder.py:207):
            # any value other than the int 1 raises there as well
            raise ValueError("harmonics_scale must be 1 (the reference's tuple*scale expression fails otherwise)")
        x = x.contiguous().float()
        f0 = f0.contiguous().float()
        n, c, lf = x.shape
        if c != schema.CONTENT_DIM or f0.shape != (n, 1, lf):
            raise ValueError(f"Decoder expects x [N,768,Lf] and f0 [N,1,Lf], got {tuple(x.shape)} / {tuple(f0.shape)}")
        if lf < 5:
            raise ValueError(f"Decoder needs at least 5 frames (reflection pad 4 on the bottleneck), got {lf}")
        lw = lf * schema.SEGMENT
        L = nat.lib()
        phi_in = None
        if isinstance(phi, torch.Tensor):
            phi_in = phi.reshape(n, schema.NUM_HARMONICS).contiguous().float()
        elif isinstance(phi, PhaseColumn):
            phi_in = phi.values
        elif phi != 0:
            phi_in = torch.full((n, schema.NUM_HARMONICS), float(phi), device=x.device)
        if phi_col is None:
            phi_col = crop[1]
        if phi_col < 0:
            phi_col += lw
        wave = out if out is not None else torch.empty(n, lw, device=x.device)         # out: e.g. a slice of the batch's output
        if tuple(wave.shape) != (n, lw) or wave.dtype != torch.float32 or not wave.is_contiguous():
            raise ValueError("Decoder: out must be a contiguous fp32 [N, 320 * Lf] tensor")
        phi_out = torch.empty(n, schema.NUM_HARMONICS, device=x.device)
        ws = self._ws.get(L.alive_decoder_workspace_bytes(n, lf), x.device)
        nat.check(L.alive_decoder_forward(self.table().array, nat.ptr(x), nat.ptr(f0), nat.ptr(phi_in), int(crop[0]),
                                          int(phi_col), n, lf, nat.ptr(wave), nat.ptr(phi_out), nat.ptr(ws), nat.stream()),
                  "alive_decoder_forward")
        return wave, PhaseColumn(phi_col, phi_out, lw)

    def forward_range(self, x, f0, f_begin):
        """Decoder.forward restricted to the frames [f_begin, f_begin + x.shape[2]) of windows whose f0 [N,1,Lf] is given
        in full (context trimming).  Frames further than the decoder's receptive field from the range edges come out
        bitwise as in forward() on the whole window; -> wave [N, 320 * frames]."""
        x = x.contiguous().float()
        f0 = f0.contiguous().float()
        n, c, nf = x.shape
        lf = f0.shape[2]
        if c != schema.CONTENT_DIM or f0.shape[:2] != (n, 1) or f_begin < 0 or f_begin + nf > lf:
            raise ValueError(f"Decoder.forward_range: x {tuple(x.shape)}, f0 {tuple(f0.shape)}, f_begin {f_begin}")
        if nf < 5:
            raise ValueError(f"Decoder needs at least 5 frames, got {nf}")
        L = nat.lib()
        wave = torch.empty(n, nf * schema.SEGMENT, device=x.device)
        ws = self._ws.get(L.alive_decoder_workspace_bytes(n, nf) + L.alive_decoder_workspace_bytes(n, lf), x.device)
        nat.check(L.alive_decoder_forward_range(self.table().array, nat.ptr(x), nat.ptr(f0), n, lf, int(f_begin), nf,
                                                nat.ptr(wave), nat.ptr(ws), nat.stream()), "alive_decoder_forward_range")
        return wave

    __call__ = forward
