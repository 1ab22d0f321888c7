"""Streaming conversion step of the reference's realtime_inference.py:122-191 as a reusable class.

int16 chunk in -> ring of `buffersize` chunks -> spectrogram / content encoder / f0 / kNN /
decoder over the whole ring (exactly what the reference recomputes per step) -> centre chunk out
as int16, with the oscillator phase carried through phi[:, :, end_of_output].
"""
import numpy as np
import torch

from . import audio_io, ops
from .common import PackedLibrary, merge_gather
from .spectrogram import spectrogram


# Interior reuse.  A content / f0 frame t of the ring depends on spectrogram frames [t - 12, t + 12] (four k7
# depthwise convs in ContentEncoder and F0Estimator) and those on samples [320 t - 640, 320 t + 640): frames at least EDGE from
# both ring edges do not see the reflect padding, so when the ring advances by a whole number of frames they are the SAME
# values one step later, s frames further left -- and so are their matched features (the kNN is per frame).  What cannot be
# carried over: the oscillator (its phase is a running sum from the ring's first sample, re-anchored every step) and with it
# the whole Filter, i.e. the decoder always runs on the full ring.
# One more condition makes the reuse EXACT rather than merely close: the library picks its kernels by problem size (below 96
# frame COLUMNS -- batch x frames -- the fp32-activation streaming kernels, from 96 the plane-packed split-bf16 GEMMs; the norm
# kernel has a per-frame form for T <= 32), and the two families round differently (1e-5).  A recomputed edge block must
# therefore go through the same kernels as the full ring would.  Blocks are 33 frames (more than 32: the norm kernel's batch
# form either way); under a ring of 96 frames or more the block is run as a BATCH OF THREE identical rows (>= 99 columns: the
# plane-packed family, whose results do not depend on the row), under a shorter ring as it is (its 33 + shift frames must stay
# below 96).  Round 2 used 96-frame blocks for long rings and had to refuse rings of 96 .. 195 frames.
EDGE = 14            # 12 frames of ConvNeXt context + 2 of STFT reflect padding
SPEC_MARGIN = 2      # STFT frames spoiled by the reflect padding of a slice
PLANES_MIN_COLS = 96  # csrc/networks.hip: use_planes()
NORM_SMALL_MAX_T = 32  # csrc/blocks.hip: alive_dwconv_norm's per-frame kernel


def reuse_block(frames, shift):
    """frames of a recomputed edge block (EDGE new frames + margin) under a ring of `frames`, or None where the ring is too
    short for two edge blocks (or, below 96 frames, where a block with the shift would reach the plane-packed family)"""
    blk = NORM_SMALL_MAX_T + 1
    ok = frames >= PLANES_MIN_COLS or blk + shift < PLANES_MIN_COLS
    return blk if ok and frames >= 2 * (blk + SPEC_MARGIN) + shift else None


def reuse_rows(frames):
    """batch rows an edge block is run with: enough identical rows to reach the kernel family of the full ring"""
    blk = NORM_SMALL_MAX_T + 1
    return 1 if frames < PLANES_MIN_COLS else -(-PLANES_MIN_COLS // blk)


class RealtimeConverter:
    def __init__(self, content_encoder, f0_estimator, decoder, library_tokens, device="cuda", chunk=960, buffersize=8,
                 input_sr=16000, output_sr=16000, f0_rate=1.0, pitch=0.0, k=4, alpha=0.0, gain=0.0, input_gain=0.0,
                 reuse_interior="auto"):
        self.device = torch.device(device)
        self.ce, self.pe, self.dec = content_encoder.to(device), f0_estimator.to(device), decoder.to(device)
        for net in (self.ce, self.pe, self.dec):
            net.table()                                # weight tables packed up front (never inside a captured step)
        self.dec._split_for_this_checkpoint()          # precision calibration of this checkpoint, also never inside a capture
        self.lib = library_tokens if isinstance(library_tokens, PackedLibrary) else PackedLibrary(library_tokens[0].to(device))
        self.chunk, self.buffersize = chunk, buffersize
        self.input_sr, self.output_sr = input_sr, output_sr
        self.f0_rate, self.pitch, self.k, self.alpha, self.gain, self.input_gain = f0_rate, pitch, k, alpha, gain, input_gain
        # realtime_inference.py:122-126
        internal_chunk = int(chunk * (16000 / output_sr))
        center = int(internal_chunk * buffersize) // 2
        self.end_of_output = center + internal_chunk // 2
        self.begin_of_output = center - internal_chunk // 2
        frames = (chunk * buffersize * 16000 // input_sr) // 320
        if frames < 5:
            raise ValueError(f"ring of {buffersize} x {chunk} samples is {frames} frames; the decoder needs >= 5 "
                             "(reflection pad 4 on the bottleneck: module/decoder.py:165 of the reference)")
        self.ring = []
        self.phi = 0
        self._graph = None
        self.last_f0 = None
        self._side = None                  # side stream of the f0 estimator (see _f0_on_side_stream)
        self._f0_bufs = {}
        # interior reuse: only where it is exact -- no resampling in front (the ring IS the 16 kHz signal), a shift of whole
        # frames, and a ring long enough that the two recomputed edge blocks do not meet.  "auto": on when that holds.
        self.frames, self.shift = frames, chunk // 320
        self._blk = reuse_block(frames, self.shift) if (input_sr == 16000 and chunk % 320 == 0 and
                                                        (chunk * buffersize) % 320 == 0) else None
        fits = self._blk is not None
        if reuse_interior is True and not fits:
            raise ValueError("interior reuse needs input_sr 16000, chunk a multiple of 320 and a ring of at least 70 + chunk / 320 "
                             f"frames (below 96 frames: 33 + chunk / 320 < 96); got {frames} frames: see module/realtime.py")
        self.reuse = bool(fits and reuse_interior in (True, "auto"))
        self._rows = reuse_rows(frames)
        self._cache_valid = False
        if self._fp16_guarded():
            ops.f16_clear()                # stale counts of earlier work in this process are not this stream's
        if self.reuse:
            self._c_feat = torch.zeros(1, 768, frames, device=self.device)      # matched features of the ring's frames
            self._c_f0 = torch.zeros(1, 1, frames, device=self.device)          # transformed f0

    # ------------------------------------------------------------------ device part of one step
    def _device_step(self, data, phi):
        """data float32 [1, ring samples at input_sr] on the device, phi [1, 64] -> (wave at output_sr [L], phi_next [1, 64])"""
        if self.reuse:
            return self._device_step_reuse(data, phi)
        data = audio_io.resample(data, self.input_sr, 16000, post_gain_db=self.input_gain)     # resample, then gain (:146-147)
        spec = spectrogram(data)
        f0, join = self._f0_on_side_stream(spec)
        content = self.ce(spec)
        val, idx = self.lib.search(content, self.k)
        content = merge_gather(val, idx, 1, self.k, self.alpha, self.lib.rows, content)
        join()
        wave, phi_out = self.dec(content, f0=f0, phi=phi, crop=(self.begin_of_output, self.end_of_output))
        self.last_f0 = f0                  # (a view of the per-shape side-stream buffer: valid until the next step)
        wave = audio_io.resample(wave, 16000, self.output_sr, pre_gain_db=self.gain)[0]         # gain, then resample (:173-175)
        return wave, phi_out[:, :, self.end_of_output]

    def _f0_on_side_stream(self, spec):
        """The f0 estimator (+ the pitch transform) needs nothing but the spectrogram and feeds nothing before the decoder: its ~35
        dependent launches run on a side stream beside the content encoder and the match (a step is a chain of ~150 small kernels,
        bound by their latencies, not by the chip).  Returns (f0, join): the f0 tensor -- a persistent buffer per shape (at most
        four shapes are kept: a converter has one ring geometry and two slice geometries), so that the steady-state step allocates
        nothing on the side stream (the estimator's scratch is sized by the eager warm-up steps that precede hipGraph capture:
        enable_graph) -- and the call that makes the current stream wait for it.  The buffer is overwritten by the next call with
        the same shape: `last_f0` of the non-reuse step aliases it and is valid until the next step.  Same kernels, same results;
        captured into the step's hipGraph as a parallel branch."""
        cur = torch.cuda.current_stream(spec.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=spec.device)
        key = (spec.shape[0], spec.shape[2])
        buf = self._f0_bufs.get(key)
        if buf is None:
            if len(self._f0_bufs) >= 4:
                self._f0_bufs.pop(next(iter(self._f0_bufs)))
            buf = self._f0_bufs[key] = torch.empty(spec.shape[0], 1, spec.shape[2], device=spec.device)
        side = self._side
        side.wait_stream(cur)                                   # the spectrogram is complete
        with torch.cuda.stream(side):
            f0 = self.pe.estimate(spec, out=buf)
            f0 = ops.pitch_transform_(f0, 1, f0_rate=self.f0_rate, pitch_shift=self.pitch)
        return f0, (lambda: cur.wait_stream(side))

    def _front_end(self, data):
        """16 kHz ring [1, n] -> (matched content [1, 768, F], transformed f0 [1, 1, F]) for every frame"""
        spec = spectrogram(data)
        f0, join = self._f0_on_side_stream(spec)
        content = self.ce(spec)
        val, idx = self.lib.search(content, self.k)
        out = merge_gather(val, idx, 1, self.k, self.alpha, self.lib.rows, content)
        join()
        return out, f0

    def _device_step_reuse(self, data, phi):
        """the same step with the front end computed for the two edge blocks only; interior frames come from the previous
        step's ring, `shift` frames further right.  Bitwise the full computation (tests/test_gpu_cli.py)."""
        F_, s = self.frames, self.shift
        data = data if self.input_gain == 0 else audio_io.gain(data, self.input_gain)
        if not self._cache_valid:
            feat, f0 = self._front_end(data)
            self._c_feat.copy_(feat)
            self._c_f0.copy_(f0)
            self._cache_valid = True
        else:
            keep_f, keep_p = self._c_feat[:, :, s:].clone(), self._c_f0[:, :, s:].clone()
            self._c_feat[:, :, :F_ - s].copy_(keep_f)
            self._c_f0[:, :, :F_ - s].copy_(keep_p)
            nl = self._blk                                            # left block: frames [0, EDGE) from a slice of nl frames
            margin = nl - EDGE                                        # >= 19 > the 12 frames of context
            feat, f0 = self._front_end_slice(data[:, :(nl + SPEC_MARGIN) * 320], 0, nl)
            self._c_feat[:, :, :EDGE].copy_(feat[:, :, :EDGE])
            self._c_f0[:, :, :EDGE].copy_(f0[:, :, :EDGE])
            a = F_ - EDGE - s                                         # right block: frames [a, F)
            lo = a - margin
            feat, f0 = self._front_end_slice(data[:, (lo - SPEC_MARGIN) * 320:], SPEC_MARGIN, SPEC_MARGIN + F_ - lo)
            self._c_feat[:, :, a:].copy_(feat[:, :, margin:])
            self._c_f0[:, :, a:].copy_(f0[:, :, margin:])
        wave, phi_out = self.dec(self._c_feat, f0=self._c_f0, phi=phi, crop=(self.begin_of_output, self.end_of_output))
        self.last_f0 = self._c_f0
        wave = audio_io.resample(wave, 16000, self.output_sr, pre_gain_db=self.gain)[0]
        return wave, phi_out[:, :, self.end_of_output]

    def _front_end_slice(self, samples, f_lo, f_hi):
        """front end on a slice of the ring: spectrogram of `samples`, frames [f_lo, f_hi) of it through the networks and the
        match (the frames outside are spoiled by the slice's own reflect padding)"""
        samples = samples.contiguous()
        if self._rows > 1:                                  # identical rows: same kernels as the full ring (see the header)
            samples = samples.expand(self._rows, -1).contiguous()
        spec = spectrogram(samples)[:, :, f_lo:f_hi].contiguous()
        f0, join = self._f0_on_side_stream(spec)
        content = self.ce(spec)
        val, idx = self.lib.search(content, self.k)
        out = merge_gather(val, idx, 1, self.k, self.alpha, self.lib.rows, content)
        join()
        return out[:1], f0[:1].clone()                   # (f0 lives in a per-shape buffer the next slice overwrites)

    def enable_graph(self):
        """Capture the whole per-step device pipeline (~150 launches) into one hipGraph: the C ABI never allocates or
        synchronises, and every scratch buffer reaches its steady-state size during the warm-up steps below."""
        n = self.chunk * self.buffersize
        self._g_in = torch.zeros(1, n, device=self.device)
        self._g_phi = torch.zeros(1, 64, device=self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._device_step(self._g_in, self._g_phi)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            wave, phi_next = self._device_step(self._g_in, self._g_phi)
            self._g_phi.copy_(phi_next)
            self._g_out = wave
        self._g_phi.zero_()
        self._cache_valid = False          # interior reuse: the captured step is the incremental one; the first real step runs in full
        return self

    def reset(self):
        """start a new stream on this converter: empty ring, phase 0, and the interior-reuse caches invalid (the next step runs
        the full front end).  Call it whenever the chunk sequence restarts or a chunk was dropped / duplicated."""
        self.ring = []
        self.phi = 0
        self._cache_valid = False
        if getattr(self, "_graph", None) is not None:
            self._g_phi.zero_()
        return self

    def step_device(self, ring_f32, continues=False):
        """ring float32 [1, buffersize*chunk] already on the device -> wave [L] (device); phase carried internally.

        Contract of interior reuse (on when the ring geometry allows, see `reuse_block`): the matched features and f0 of the
        ring's interior frames are carried over from the previous call, which is only right when `ring_f32` IS the previous
        ring advanced by exactly one chunk.  The caller says so with `continues=True` (`step()` does: it owns the ring);
        the default treats the ring as unrelated to the previous one and recomputes the whole front end."""
        if not continues:
            self._cache_valid = False
        if getattr(self, "_graph", None) is not None:
            if self.reuse and not self._cache_valid:             # fills the frame caches the captured (incremental) step reads
                wave, phi_next = self._device_step(ring_f32, self._g_phi)
                self._g_phi.copy_(phi_next)
                return wave
            self._g_in.copy_(ring_f32)
            self._graph.replay()
            return self._g_out
        phi = self.phi if isinstance(self.phi, torch.Tensor) else torch.zeros(1, 64, device=self.device)
        wave, phi_next = self._device_step(ring_f32, phi)
        self.phi = phi_next
        return wave

    def _fp16_guarded(self):
        """Rings of 96 frames or more run the batch kernels and with them the fp16 forms of the encoder / decoder GEMMs (modes 1):
        `step()` then reads the saturation counters after every chunk (it has just synchronised for the PCM copy).  Shorter rings --
        the reference's defaults -- run the fp32-activation streaming kernels, which write no fp16 plane: nothing to guard."""
        return self.frames >= PLANES_MIN_COLS and (ops.encoder_precision(0) != 2 or ops.decoder_precision(0) != 2)

    def _repeat_on_bf16(self, data, saved_phi):
        """a chunk drove an activation out of fp16's range: switch the process to precision modes 2 (bf16 planes, fp32's range -- a
        stream that saturates once will do so again, so the modes stay), restore the phase the chunk started from, drop the frame
        caches, re-capture the step if it was a hipGraph, and convert the chunk again"""
        import warnings
        warnings.warn("an activation left fp16's range in the streaming step: switching to ALIVE_ENCODER_PRECISION=2 / "
                      "ALIVE_DECODER_PRECISION=2 (bf16 planes) and converting the chunk again", RuntimeWarning)
        ops.Fp16Guard.fallbacks += 1
        ops.encoder_precision(2)
        ops.decoder_precision(2)
        self._cache_valid = False
        if getattr(self, "_graph", None) is not None:
            self.enable_graph()
            self._g_phi.copy_(saved_phi)
        else:
            self.phi = saved_phi
        wave = self.step_device(data, continues=False)
        return audio_io.float_to_pcm16(wave).cpu().numpy()

    def step(self, data_int16: np.ndarray):
        """one chunk of int16 samples -> converted centre chunk (int16), or None while the ring fills
        (the reference's loop emits nothing until it holds more than `buffersize` chunks: :133-137)."""
        self.ring.append(np.asarray(data_int16, dtype=np.int16))
        if len(self.ring) > self.buffersize:
            del self.ring[0]
        else:
            return None
        data = torch.from_numpy(np.concatenate(self.ring, 0)).to(self.device)
        data = audio_io.pcm16_to_float(data).unsqueeze(0)            # / 32768 on the device (:139-140)
        guarded = self._fp16_guarded()
        if guarded:
            saved_phi = self._g_phi.clone() if getattr(self, "_graph", None) is not None else self.phi
        wave = self.step_device(data, continues=True)                # this ring is the previous one advanced by one chunk
        out = audio_io.float_to_pcm16(wave).cpu().numpy()            # C cast of numpy's astype, no clipping (:180-183)
        if guarded and ops.f16_saturations(reset=True) > 0:          # (the copy above has synchronised: five 4-byte reads)
            out = self._repeat_on_bf16(data, saved_phi)
        center = self.buffersize * self.chunk // 2
        return out[center - self.chunk // 2: center + self.chunk // 2]
