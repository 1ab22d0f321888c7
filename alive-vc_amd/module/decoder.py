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


# Round 6: decoder precision mode 1 (plain fp16 operands in five layer groups, csrc/networks.hip) was chosen on ONE weight set.  On the
# reference fixtures of two more seeds it behaves the same (4.4e-5 .. 5.8e-5 of the waveform RMS), but on a set whose FiLM projections and
# pointwise convs carry 4 x the weights of a fresh initialisation every group is 15 - 25 x more sensitive (the Filter has no normalisation,
# FiLM gains multiply activations: /root/reference/module/decoder.py:112-117,153-195) and the mode reaches 8.6e-4 of the waveform RMS --
# against 2.2e-5 for split bf16 (profiles/r06_decoder_precision_groups.json).  So the fp16 mode is CALIBRATED PER CHECKPOINT: when a
# decoder is first used (and nobody has chosen a mode through ALIVE_DECODER_PRECISION or `Decoder.precision`), a 128-frame seeded probe
# is decoded in both modes, and the checkpoint keeps the fp16 forms only if they stay within CALIBRATION_BAR of the split-bf16 result,
# relative to the probe waveform's RMS.  Cost: two decodes of 128 frames per checkpoint load.
CALIBRATION_BAR = 1.5e-4
CALIBRATION_FRAMES = 128


class Decoder(PackedNet):
    MODEL_ID = 2
    PREFIX = "dec."
    _schema = staticmethod(schema.decoder_schema)
    _pack = staticmethod(pack_decoder)
    # None: calibrated at first use (unless ALIVE_DECODER_PRECISION is set); 1: the fp16 forms whenever the process mode allows them;
    # 2: split bf16 for this checkpoint whatever the process mode
    precision = None
    calibration = None                      # {"relative_difference": ..., "chosen": 1 | 2} once calibrated

    def load_state_dict(self, sd, strict=True):
        self.calibration = None
        return super().load_state_dict(sd, strict)

    def _calibrate(self):
        import os
        import warnings
        from . import ops, synthetic
        if self.calibration is not None or self.precision is not None or os.environ.get("ALIVE_DECODER_PRECISION") or ops.decoder_precision(0) != 1:
            return
        if torch.cuda.is_current_stream_capturing():         # (Converter / RealtimeConverter calibrate when they are built: never in a capture)
            return
        self.calibration = {"chosen": 1}                     # (recursion guard: the probe goes through forward)
        lf = CALIBRATION_FRAMES
        x = synthetic.gaussian("decoder.calibration.x", 1, (1, schema.CONTENT_DIM, lf)).to(self._device)
        t = torch.arange(lf, dtype=torch.float32)
        f0 = (140.0 * (1.0 + 0.3 * torch.sin(2 * torch.pi * t / 57.0))).view(1, 1, lf).clone()
        f0[:, :, 40:46] = 0.0
        f0 = f0.to(self._device)
        ops.f16_clear()
        try:
            w1 = self.forward(x, f0)[0]
            sat = ops.f16_saturations(reset=True)             # (synchronises) a probe that leaves fp16's range settles it at once
            ops.decoder_precision(2)
            w2 = self.forward(x, f0)[0]
        finally:
            ops.decoder_precision(1)
        rel = float(((w1.double() - w2.double()).pow(2).mean().sqrt() / w2.double().pow(2).mean().sqrt().clamp(min=1e-30)).item())
        ok = rel == rel and rel <= CALIBRATION_BAR and sat == 0
        self.calibration = {"relative_difference": rel, "bar": CALIBRATION_BAR, "fp16_saturations": sat, "chosen": 1 if ok else 2}
        if not ok:
            warnings.warn(f"decoder precision calibration: the fp16 forms differ from split bf16 by {rel:.2e} of the probe waveform's RMS "
                          f"(bar {CALIBRATION_BAR:.1e}) on this checkpoint: it runs on split bf16 (ALIVE_DECODER_PRECISION=1 forces fp16)",
                          RuntimeWarning)

    def _split_for_this_checkpoint(self):
        if self.precision is not None:
            return self.precision == 2
        if self.calibration is None and self._device.type == "cuda":
            self._calibrate()
        return self.calibration is not None and self.calibration.get("chosen") == 2

    def _run(self, call):
        """`call()` (one C-ABI decoder entry point) in this checkpoint's precision: the process mode is switched to 2 around it when the
        checkpoint needs split bf16 (the mode is read on the host while the kernels are enqueued: nothing asynchronous depends on it)"""
        from . import ops
        if self._split_for_this_checkpoint() and ops.decoder_precision(0) == 1:
            ops.decoder_precision(2)
            try:
                return call()
            finally:
                ops.decoder_precision(1)
        return call()

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
        tab = self.table()
        self._run(lambda: nat.check(L.alive_decoder_forward(tab.array, nat.ptr(x), nat.ptr(f0), nat.ptr(phi_in), int(crop[0]),
                                                            int(phi_col), n, lf, nat.ptr(wave), nat.ptr(phi_out), nat.ptr(ws), nat.stream()),
                                    "alive_decoder_forward"))
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
        tab = self.table()
        self._run(lambda: nat.check(L.alive_decoder_forward_range(tab.array, nat.ptr(x), nat.ptr(f0), n, lf, int(f_begin), nf,
                                                                  nat.ptr(wave), nat.ptr(ws), nat.stream()), "alive_decoder_forward_range"))
        return wave

    __call__ = forward
