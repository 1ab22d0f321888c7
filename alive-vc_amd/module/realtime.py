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


class RealtimeConverter:
    def __init__(self, content_encoder, f0_estimator, decoder, library_tokens, device="cuda", chunk=960, buffersize=8,
                 input_sr=16000, output_sr=16000, f0_rate=1.0, pitch=0.0, k=4, alpha=0.0, gain=0.0, input_gain=0.0):
        self.device = torch.device(device)
        self.ce, self.pe, self.dec = content_encoder.to(device), f0_estimator.to(device), decoder.to(device)
        for net in (self.ce, self.pe, self.dec):
            net.table()                                # weight tables packed up front (never inside a captured step)
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

    # ------------------------------------------------------------------ device part of one step
    def _device_step(self, data, phi):
        """data float32 [1, ring samples at input_sr] on the device, phi [1, 64] -> (wave at output_sr [L], phi_next [1, 64])"""
        data = audio_io.resample(data, self.input_sr, 16000, post_gain_db=self.input_gain)     # resample, then gain (:146-147)
        spec = spectrogram(data)
        content = self.ce(spec)
        f0 = self.pe.estimate(spec)
        f0 = ops.pitch_transform_(f0, 1, f0_rate=self.f0_rate, pitch_shift=self.pitch)
        val, idx = self.lib.search(content, self.k)
        content = merge_gather(val, idx, 1, self.k, self.alpha, self.lib.rows, content)
        wave, phi_out = self.dec(content, f0=f0, phi=phi, crop=(self.begin_of_output, self.end_of_output))
        self.last_f0 = f0
        wave = audio_io.resample(wave, 16000, self.output_sr, pre_gain_db=self.gain)[0]         # gain, then resample (:173-175)
        return wave, phi_out[:, :, self.end_of_output]

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
        return self

    def step_device(self, ring_f32):
        """ring float32 [1, buffersize*chunk] already on the device -> wave [L] (device); phase carried internally."""
        if getattr(self, "_graph", None) is not None:
            self._g_in.copy_(ring_f32)
            self._graph.replay()
            return self._g_out
        phi = self.phi if isinstance(self.phi, torch.Tensor) else torch.zeros(1, 64, device=self.device)
        wave, phi_next = self._device_step(ring_f32, phi)
        self.phi = phi_next
        return wave

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
        wave = self.step_device(data)
        out = audio_io.float_to_pcm16(wave).cpu().numpy()            # C cast of numpy's astype, no clipping (:180-183)
        center = self.buffersize * self.chunk // 2
        return out[center - self.chunk // 2: center + self.chunk // 2]
