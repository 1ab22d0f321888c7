"""The conversion loop of the reference's inference.py:94-135 as a batched device pipeline.

The reference walks the overlapping windows of one utterance serially (N = 1, two host syncs per
window).  Windows are independent units (each has its own mean pitch, oscillator phase origin
and reflect pads), so here all windows of all utterances go through the networks as one batch
[n_windows, ...]; results are identical to per-window calls.
"""
import os

import torch
import torch.nn.functional as F

from . import ops
from .common import PackedLibrary, merge_gather
from .content_encoder import ContentEncoder
from .decoder import Decoder
from .f0_estimator import F0Estimator
from .spectrogram import spectrogram


def make_windows(wf, chunk=48000):
    """inference.py:94-101: wf [1, L] -> windows [n, 3*chunk] (hop = chunk), total length."""
    total = wf.shape[1]
    wf = torch.cat([wf, torch.zeros(1, chunk * 3, device=wf.device, dtype=wf.dtype)], dim=1)
    wf = F.pad(wf, (chunk, chunk))
    n = (wf.shape[1] - 3 * chunk) // chunk + 1
    return wf.unfold(1, 3 * chunk, chunk)[0, :n].contiguous(), total


def stitch(window_out, total, chunk=48000):
    """inference.py:132-135: keep the centre third of every window, concatenate, trim."""
    return window_out[:, chunk:2 * chunk].reshape(1, -1)[:, :total]


# Frames of the matched features that can reach an output frame t through the decoder (module/decoder.py of the
# reference): the FeatureExtractor is four k7 depthwise convs (+-12 frames); the Filter is causal -- mid conv 4 frames,
# the 256 / 64 / 16 / 8-channel blocks 56 columns each at 32 / 4 / 2 / 1 samples (6.7 frames together) -- and reads FiLM
# / amplitudes interpolated from the neighbouring frames (+-1); source_in / source_out add 3 samples either side.
# => features [t - 24, t + 13].  The margins below leave slack and are checked bitwise by the tests.
TRIM_LEFT, TRIM_RIGHT = 32, 16
CE_MARGIN = 16          # ContentEncoder: four k7 depthwise convs (+-12 frames) between the spectrogram and its output


# Overlap sharing.  Consecutive windows of one utterance overlap by two chunks, and a content / f0 frame at least EDGE frames
# from both edges of its window depends on nothing but the samples around it (spectrogram frame +-640 samples, four k7
# depthwise convs +-12 frames): it has the same value in every window that contains it, and so has its kNN match.  With
# `share_overlap` the front end (spectrogram, f0 estimator, content encoder, match) therefore runs ONCE per utterance on the
# padded signal plus, per window, on its two edge blocks; every window's matched features and raw f0 are then assembled by
# copies.  The pitch transform (per-window mean pitch) and the whole decoder still run per window.  Results are bitwise those
# of the per-window front end (tests/test_gpu_batch.py); at the reference's 3-chunk windows it launches ~51 % of the frames.
EDGE = 14            # 12 frames of ConvNeXt context + 2 of STFT reflect padding (module/realtime.py uses the same numbers)
NET_MARGIN = 16
SPEC_MARGIN = 2


def _edge_chunks(n, size):
    """[i0, i1) bounds of the edge-block launches over n windows, `size` windows at a time.  An edge block is EDGE + NET_MARGIN = 30
    frame columns per window, and below 96 columns the library switches to its fp32 streaming kernels, which round differently from
    the plane GEMMs every other window's edge blocks run on: a tail of fewer than 4 windows is folded into the chunk before it, so
    that the stitched output stays bitwise that of the per-window path for every window count."""
    need = -(-96 // (EDGE + NET_MARGIN))
    bounds = [(i, min(n, i + size)) for i in range(0, n, size)]
    if len(bounds) > 1 and bounds[-1][1] - bounds[-1][0] < need:
        last = bounds.pop()
        bounds[-1] = (bounds[-1][0], last[1])
    return bounds


class Converter:
    def __init__(self, content_encoder: ContentEncoder, f0_estimator: F0Estimator, decoder: Decoder, device="cuda"):
        self.device = torch.device(device)
        self.ce, self.pe, self.dec = content_encoder.to(device), f0_estimator.to(device), decoder.to(device)
        self.library = None
        for net in (self.ce, self.pe, self.dec):       # pack the weight tables now, on the caller's stream (not lazily
            net.table()                                # inside the first window batch, which runs on a side stream)
        if self.device.type == "cuda":
            self.dec._split_for_this_checkpoint()      # ... and calibrate the decoder's precision mode for this checkpoint (module/decoder.py)

    def set_library(self, tokens):
        """tokens [1, 768, M] (voice_library.pt layout) or an already packed PackedLibrary."""
        self.library = tokens if isinstance(tokens, PackedLibrary) else PackedLibrary(tokens[0].to(self.device))
        return self

    def features(self, windows, pitch_shift=0.0, intonation=1.0, f0_rate=1.0, frames=None, out=None):
        """spectrogram -> f0 (+ per-window pitch transform) and content features   (inference.py:112-128).
        frames = (lo, hi): content features are only needed on [lo, hi) (context trimming): the encoder runs on that range
        plus its own receptive field; the other frames come back as zeros.  f0 always covers the whole window (the pitch
        transform uses the window's mean pitch, the oscillator accumulates phase from the first frame)."""
        if frames is None:                 # the whole window: one fused call, no fp32 spectrogram (ops.front_end)
            feat, f0 = ops.front_end(windows, self.ce, self.pe, out=out)     # out = (feat, f0) batch slices to write into
            return feat, ops.pitch_transform_(f0, 0, f0_rate=f0_rate, pitch_shift=pitch_shift, intonation=intonation)
        spec = spectrogram(windows)
        f0 = self.pe.estimate(spec, out=None if out is None else out[1])
        f0 = ops.pitch_transform_(f0, 0, f0_rate=f0_rate, pitch_shift=pitch_shift, intonation=intonation)
        lf = spec.shape[2]
        a, b = max(0, frames[0] - CE_MARGIN), min(lf, frames[1] + CE_MARGIN)
        feat = torch.zeros(spec.shape[0], 768, lf, device=spec.device)
        feat[:, :, a:b] = self.ce(spec[:, :, a:b].contiguous())
        return feat, f0

    def features_shared(self, windows, group, k=4, alpha=0.0, utt_batch=32, frames=None):
        """(matched content features [n, 768, lf], raw f0 [n, 1, lf]) of n = m * group windows, every `group` consecutive ones
        being the windows of one signal at hop = a third of the window (make_windows): the front end runs once per signal.
        frames = (r0, r1) with EDGE <= r0 < r1 <= lf - EDGE (context trimming on top of sharing): only the matched features of
        frames [r0, r1) of every window are needed -> [n, 768, r1 - r0].  They are all interior frames of the signal, so the
        edge blocks carry the f0 estimator alone (f0 still covers the whole window) and the match runs on the union of the
        windows' ranges, frames [r0, (group - 1) cf + r1) of each signal."""
        n, L = windows.shape
        c = L // 3
        lf, cf = L // 320, c // 320
        if n % group or c % 320 or lf < 2 * (EDGE + NET_MARGIN + SPEC_MARGIN):
            raise ValueError(f"share_overlap needs groups of {group} windows of 3 chunks, chunk a multiple of 320 and >= "
                             f"{2 * (EDGE + NET_MARGIN + SPEC_MARGIN)} frames per window")
        if frames is not None:
            return self._features_shared_range(windows, group, k, alpha, utt_batch, frames)
        m = n // group
        w3 = windows.view(m, group, L)
        # the padded signal of every group: first chunk of each window, then the last two chunks of the last one
        sig = torch.cat([w3[:, :, :c].reshape(m, group * c), w3[:, -1, c:]], dim=1).contiguous()      # [m, (group + 2) c]
        dev = windows.device
        feat = torch.empty(n, 768, lf, device=dev)
        f0 = torch.empty(n, 1, lf, device=dev)
        fu = torch.empty(m, 768, (group + 2) * cf, device=dev)
        pu = torch.empty(m, 1, (group + 2) * cf, device=dev)
        for i in range(0, m, utt_batch):                       # interior frames: one pass over each signal
            ops.front_end(sig[i:i + utt_batch], self.ce, self.pe, out=(fu[i:i + utt_batch], pu[i:i + utt_batch]))
        nl = EDGE + NET_MARGIN
        fe = torch.empty(n, 768, 2 * EDGE, device=dev)         # edge frames of every window: [0, EDGE) and [lf - EDGE, lf)
        pe_ = torch.empty(n, 1, 2 * EDGE, device=dev)
        for i, j in _edge_chunks(n, 8 * utt_batch):
            w = windows[i:j]
            sl = spectrogram(w[:, :(nl + SPEC_MARGIN) * 320].contiguous())[:, :, :nl].contiguous()
            sr = spectrogram(w[:, L - (nl + SPEC_MARGIN) * 320:].contiguous())[:, :, SPEC_MARGIN:].contiguous()
            fe[i:j, :, :EDGE] = self.ce(sl)[:, :, :EDGE]
            fe[i:j, :, EDGE:] = self.ce(sr)[:, :, NET_MARGIN:]
            pe_[i:j, :, :EDGE] = self.pe.estimate(sl)[:, :, :EDGE]
            pe_[i:j, :, EDGE:] = self.pe.estimate(sr)[:, :, NET_MARGIN:]
        # the match: once per distinct frame, and ONE search for the signals' frames and the windows' edge frames together
        # (a search of its own for the 28 edge frames per window would run the scoring kernel at a fraction of its rate)
        tu = fu.shape[2]
        src = torch.cat([fu.permute(1, 0, 2).reshape(768, m * tu), fe.permute(1, 0, 2).reshape(768, n * 2 * EDGE)], dim=1)
        matched = self.match(src.unsqueeze(0).contiguous(), k, alpha)[0]
        fu = matched[:, :m * tu].view(768, m, tu).permute(1, 0, 2)
        fe = matched[:, m * tu:].view(768, n, 2 * EDGE).permute(1, 0, 2)
        f4, p4 = feat.view(m, group, 768, lf), f0.view(m, group, 1, lf)
        for g in range(group):                                 # window g of a signal = frames [g cf, g cf + lf) of it
            f4[:, g, :, EDGE:lf - EDGE] = fu[:, :, g * cf + EDGE:g * cf + lf - EDGE]
            p4[:, g, :, EDGE:lf - EDGE] = pu[:, :, g * cf + EDGE:g * cf + lf - EDGE]
        feat[:, :, :EDGE], feat[:, :, lf - EDGE:] = fe[:, :, :EDGE], fe[:, :, EDGE:]
        f0[:, :, :EDGE], f0[:, :, lf - EDGE:] = pe_[:, :, :EDGE], pe_[:, :, EDGE:]
        self.last_front_end_frames = m * (group + 2) * cf + n * 2 * EDGE           # frames that went through the match
        return feat, f0

    def _features_shared_range(self, windows, group, k, alpha, utt_batch, frames):
        n, L = windows.shape
        c = L // 3
        lf, cf = L // 320, c // 320
        r0, r1 = frames
        if not (EDGE <= r0 < r1 <= lf - EDGE):
            raise ValueError(f"shared front end with a frame range needs {EDGE} <= r0 < r1 <= {lf - EDGE}, got {frames}")
        m = n // group
        nr = r1 - r0
        w3 = windows.view(m, group, L)
        sig = torch.cat([w3[:, :, :c].reshape(m, group * c), w3[:, -1, c:]], dim=1).contiguous()      # [m, (group + 2) c]
        dev = windows.device
        tu = (group + 2) * cf
        u0, u1 = r0, (group - 1) * cf + r1                      # frames of a signal that some window's range holds
        a, b = max(0, u0 - CE_MARGIN), min(tu, u1 + CE_MARGIN)  # ... plus the content encoder's own receptive field
        pu = torch.empty(m, 1, tu, device=dev)
        fu = torch.empty(m, 768, u1 - u0, device=dev)
        for i in range(0, m, utt_batch):
            spec = spectrogram(sig[i:i + utt_batch])
            pu[i:i + utt_batch] = self.pe.estimate(spec)
            fu[i:i + utt_batch] = self.ce(spec[:, :, a:b].contiguous())[:, :, u0 - a:u1 - a]
        nl = EDGE + NET_MARGIN
        f0 = torch.empty(n, 1, lf, device=dev)
        for i, j in _edge_chunks(n, 8 * utt_batch):            # f0 of the edge frames of every window
            w = windows[i:j]
            sl = spectrogram(w[:, :(nl + SPEC_MARGIN) * 320].contiguous())[:, :, :nl].contiguous()
            sr = spectrogram(w[:, L - (nl + SPEC_MARGIN) * 320:].contiguous())[:, :, SPEC_MARGIN:].contiguous()
            f0[i:j, :, :EDGE] = self.pe.estimate(sl)[:, :, :EDGE]
            f0[i:j, :, lf - EDGE:] = self.pe.estimate(sr)[:, :, NET_MARGIN:]
        nu = u1 - u0
        matched = self.match(fu.permute(1, 0, 2).reshape(1, 768, m * nu).contiguous(), k, alpha)[0].view(768, m, nu).permute(1, 0, 2)
        feat = torch.empty(n, 768, nr, device=dev)
        f4, p4 = feat.view(m, group, 768, nr), f0.view(m, group, 1, lf)
        for g in range(group):                                 # window g of a signal = frames [g cf, g cf + lf) of it
            f4[:, g] = matched[:, :, g * cf:g * cf + nr]
            p4[:, g, :, EDGE:lf - EDGE] = pu[:, :, g * cf + EDGE:g * cf + lf - EDGE]
        self.last_front_end_frames = m * nu                     # frames that went through the match
        return feat, f0

    def match(self, feat, k=4, alpha=0.0):
        val, idx = self.library.search(feat, k)
        return merge_gather(val, idx, 1, k, alpha, self.library.rows, feat)

    def convert_windows(self, windows, *a, **kw):
        """`_convert_windows` (below: same arguments) under the fp16 range guard (ops.Fp16Guard): the saturation counters of the fp16
        forms are cleared in stream order when the batch starts and read when it ends (one device synchronisation per call); a batch
        that drove an activation out of fp16's range is REPEATED on bf16 planes instead of being returned saturated -- every public
        batch entry point (this one, `convert`, `generate_voice_library.py`, `RealtimeConverter` above 96 columns) is covered."""
        return ops.Fp16Guard(self._agree_on_saturations()).run(lambda: self._convert_windows(windows, *a, **kw))

    def _agree_on_saturations(self):
        return None                       # one process, one decision (ShardedConverter: max over the ranks)

    def _convert_windows(self, windows, k=4, alpha=0.0, pitch_shift=0.0, intonation=1.0, f0_rate=1.0, window_batch=64,
                         keep_frames=None, share_overlap=None):
        """windows [n, L] on the device -> waveforms [n, L]; L a multiple of 320.
        Networks run in batches of `window_batch` windows (bounded scratch); the kNN match runs ONCE over the frames
        of all windows, so every library tile streamed from L2 is used by as many frames as possible.
        keep_frames = (a, b): the caller keeps only the output of frames [a, b) of every window (inference.py keeps the
        centre third).  The match is then restricted to the frames that can reach them through the decoder; the kept
        samples are bitwise the same as without it (the other frames of the window decode from unmatched features)."""
        n, L = windows.shape
        lf = L // 320
        # (the edge blocks of all windows form one launch of n x 30 frame columns: below 96 columns the library would switch
        # to its streaming kernels, which round differently from the plane GEMMs the windows themselves run on -- then the
        # per-window front end is used, so that the result never depends on the flag)
        rng = None if keep_frames is None else (max(0, keep_frames[0] - TRIM_LEFT), min(lf, keep_frames[1] + TRIM_RIGHT))
        # sharing and trimming together need the trimmed range to consist of interior frames of the signal (it does for the
        # centre third of a 3-chunk window from 46 frames per chunk on); otherwise trimming alone is used -- same samples
        share_ok = bool(share_overlap) and n * (EDGE + NET_MARGIN) >= 96
        if share_ok and (rng is None or (EDGE <= rng[0] and rng[1] <= lf - EDGE and rng[1] - rng[0] >= 5)):
            # share_overlap = windows per signal (make_windows order): front end once per signal, decoder per window
            feat, f0 = self.features_shared(windows, int(share_overlap), k, alpha, frames=rng)
            f0 = ops.pitch_transform_(f0, 0, f0_rate=f0_rate, pitch_shift=pitch_shift, intonation=intonation)
            out = torch.empty_like(windows) if rng is None else torch.zeros_like(windows)
            cur = torch.cuda.current_stream()
            side = self._side_streams(windows.device)

            def dec_shared(i):
                if rng is None:
                    self.dec(feat[i:i + window_batch], f0[i:i + window_batch], out=out[i:i + window_batch])
                else:       # feat holds frames [rng[0], rng[1]) only; samples outside the range are not kept by the caller
                    out[i:i + window_batch, rng[0] * 320:rng[1] * 320] = self.dec.forward_range(
                        feat[i:i + window_batch], f0[i:i + window_batch], rng[0])
            for j, i in enumerate(range(0, n, window_batch)):
                st = side[j % len(side)] if side else None
                if st is None:
                    dec_shared(i)
                    continue
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    dec_shared(i)
            for st in side:
                cur.wait_stream(st)
            return out
        feat = torch.empty(n, 768, lf, device=windows.device)
        f0 = torch.empty(n, 1, lf, device=windows.device)

        # Window batches are independent before and after the match: they go round-robin onto side streams (scratch is per
        # stream, module/_native.py::Workspace), so that the tail of one batch's kernels overlaps the next batch's -- the
        # kernels of a batch differ widely in what bounds them.  Same kernels on the same data: results are unchanged.
        def batches(fn):
            cur = torch.cuda.current_stream()
            side = self._side_streams(windows.device)
            for j, i in enumerate(range(0, n, window_batch)):
                if not side:
                    fn(i)
                    continue
                st = side[j % len(side)]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    fn(i)
            for st in side:
                cur.wait_stream(st)

        def enc(i):
            if rng is None:                                 # the networks write straight into the batch's slices
                self.features(windows[i:i + window_batch], pitch_shift, intonation, f0_rate,
                              out=(feat[i:i + window_batch], f0[i:i + window_batch]))
                return
            feat[i:i + window_batch], f0[i:i + window_batch] = self.features(windows[i:i + window_batch], pitch_shift,
                                                                           intonation, f0_rate, frames=rng)
        batches(enc)
        if rng is None:
            feat = self.match(feat, k, alpha)
        else:
            feat[:, :, rng[0]:rng[1]] = self.match(feat[:, :, rng[0]:rng[1]].contiguous(), k, alpha)
        if rng is None:
            out = torch.empty_like(windows)

            def dec(i):
                self.dec(feat[i:i + window_batch], f0[i:i + window_batch], out=out[i:i + window_batch])
            batches(dec)
            return out
        # decode the matched range only (the oscillator still accumulates phase over the whole window); samples outside it
        # are not kept by the caller and stay zero
        out = torch.zeros_like(windows)

        def dec_range(i):
            out[i:i + window_batch, rng[0] * 320:rng[1] * 320] = self.dec.forward_range(
                feat[i:i + window_batch, :, rng[0]:rng[1]].contiguous(), f0[i:i + window_batch], rng[0])
        batches(dec_range)
        return out

    def _side_streams(self, device):
        """side streams of the window batches (ALIVE_STREAMS, default 3; 1 = everything on the caller's stream)"""
        want = int(os.environ.get("ALIVE_STREAMS", "3"))
        if want <= 1:
            return []
        pool = getattr(self, "_streams", None)
        if pool is None or len(pool) != want:
            pool = self._streams = [torch.cuda.Stream(device=device) for _ in range(want)]
        return pool

    def convert(self, wf, chunk=48000, trim_context=False, **kw):
        """one utterance: wf [1, L] at 16 kHz (already normalised / mono) -> [1, L].
        trim_context: match and decode only the frames that can reach the kept centre third (same samples, ~44 % of the kNN
        and decoder work); share_overlap=True / "auto": the front end once per utterance.  The two combine (the CLI's default):
        the stitched output is bitwise that of the plain per-window path."""
        windows, total = make_windows(wf.to(self.device), chunk)
        keep = (chunk // 320, 2 * chunk // 320) if trim_context else None
        share = kw.get("share_overlap")
        if share is True or share == "auto":                      # one utterance: all of its windows form one group
            n = windows.shape[0]
            ok = chunk % 320 == 0 and 3 * chunk // 320 >= 2 * (EDGE + NET_MARGIN + SPEC_MARGIN)
            # "auto": sharing adds ~2.5 ms of small launches (one more encoder pass, edge blocks, copies) and saves about half
            # of the front end, most of it in the kNN match -- measured break-even: 43 windows at a 50 k-vector library, 18 at 1 M
            if share == "auto" and self.library is not None:
                ok = ok and n * (1.0 + self.library.M / 5e5) >= 45.0
            kw["share_overlap"] = n if ok else None
        return stitch(self.convert_windows(windows, keep_frames=keep, **kw), total, chunk)       # (the guard sits in convert_windows)

    @staticmethod
    def check_fp16_range():
        """For callers of the LOW-LEVEL ops (ContentEncoder / Decoder objects called directly): raises when a value left fp16's range
        since the counters were last cleared, and clears them.  `convert_windows` / `convert` do not need it: they run under
        ops.Fp16Guard, which repeats a saturated batch on bf16 planes.  Synchronises the device."""
        from . import ops
        n = ops.f16_saturations(reset=True)
        if n > 0:
            raise RuntimeError(f"{n} activation value(s) left fp16's range and were saturated in the fp16 forms of the encoder / decoder GEMMs: "
                               "this checkpoint (or input scale) needs ALIVE_ENCODER_PRECISION=2 and ALIVE_DECODER_PRECISION=2 "
                               "(module.ops.encoder_precision(2), decoder_precision(2)): bf16 planes keep fp32's range")
