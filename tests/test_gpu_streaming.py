"""Long-horizon parity of the streaming loop (BASELINE config 5: `-c 160 -b 16`, the per-step device pipeline replayed from
one hipGraph) against the CPU oracle with the oscillator phase carried from step to step through asin(sin(theta))
(realtime_inference.py:166-167, decoder.py:91-95).  VERDICT r2 item 5: hundreds of steps, the WHOLE emitted stream compared,
and the drift of the carried phase measured -- the phase is re-derived every step from a rounded sine, so nothing but this
test shows that the two implementations do not walk apart."""
import json
import os

import numpy as np
import pytest
import torch

import alive_oracle as O
from module import schema, synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_streaming_300_steps_hipgraph_matches_oracle_with_carried_phase():
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.realtime import RealtimeConverter
    chunk, bs, steps = 160, 16, 300                              # 10 ms chunks, ring of 8 frames
    lib = synthetic.make_library(1000, 1)
    ce, pe, dec = (synthetic.make_state_dict(s, 2, p) for s, p in ((schema.content_encoder_schema(), "ce."),
                                                                  (schema.f0_estimator_schema(), "pe."),
                                                                  (schema.decoder_schema(), "dec.")))
    rt = RealtimeConverter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), lib, "cuda", chunk=chunk,
                           buffersize=bs, f0_rate=0.5).enable_graph()
    pcm = (synthetic.make_waveform(chunk * (bs + steps), 67)[0].numpy() * 20000).astype(np.int16)
    begin, end = O.realtime_geometry(chunk, bs)
    c = bs * chunk // 2
    phi = 0
    got, want, drift, f0_flips, resync = [], [], [], 0, 0
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for s in range(bs + steps):
        o = rt.step(pcm[s * chunk:(s + 1) * chunk])
        if s < bs:
            assert o is None
            continue
        ring = torch.from_numpy(pcm[(s - bs + 1) * chunk:(s + 1) * chunk].astype(np.float32) / 32768)[None]
        # the oracle's step, spelled out (O.realtime_step) so that the f0 of the step can be compared too
        spec = O.spectrogram(ring)
        content = O.content_encoder(ce, spec)
        f0 = O.pitch_transform_realtime(O.f0_estimate(pe, spec) * 0.5, 0.0)
        content = O.match_features(content, lib, k=4, alpha=0.0)
        wave, phi_out = O.decoder(dec, content, f0, phi=phi, crop0=begin)
        phi = phi_out[:, :, end].unsqueeze(2)
        dev_phi = rt._g_phi.detach().cpu().view(1, 64, 1)
        # An argmax flip of the f0 estimator between the two implementations (a near-tie of two logits; its own parity is
        # tests/test_gpu_models.py) would put the two phase tracks on different frequencies from here on: that is not
        # what this test measures, so the oracle is re-anchored on the device's phase and the step is not compared.
        if not torch.allclose(rt.last_f0.cpu(), f0, rtol=1e-5, atol=1e-3):      # (the pitch transform itself may differ by an ulp)
            f0_flips += 1
            resync += 1
            phi = dev_phi.clone()
            continue
        # drift of the carried phase: asin is ill-conditioned where |sin| -> 1 (d asin = d sin / cos); compare away from the fold
        ok = phi.abs() < 1.45                                        # |sin theta| < 0.9927
        d = (dev_phi - phi).abs()[ok]
        drift.append(float(d.max()) if d.numel() else 0.0)
        got.append(o.astype(np.float64))
        want.append((wave[0].numpy() * 32768).astype(np.int16)[c - chunk // 2: c + chunk // 2].astype(np.float64))
    got, want = np.concatenate(got), np.concatenate(want)
    rms = float(np.sqrt(np.mean((got - want) ** 2)) / 32768)
    drift = np.array(drift)
    report = {"steps_compared": int(drift.size), "f0_argmax_flips_resynced": f0_flips, "int16_rms_over_32768": rms,
              "phase_drift_rad_max": float(drift.max()), "phase_drift_rad_first_50_max": float(drift[:50].max()),
              "phase_drift_rad_last_50_max": float(drift[-50:].max()), "phase_drift_rad_mean": float(drift.mean())}
    print("streaming long horizon:", json.dumps(report))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(report, open(os.path.join(out_dir, "streaming_long_horizon.json"), "w"), indent=1)
    assert resync <= 3, report                                       # the comparison must cover (nearly) the whole stream
    assert drift.size >= steps - 3
    assert rms < 1e-4, report                                        # the whole emitted stream (bar 1e-3; measured 3.8e-6)
    # the carried phase does not walk away (measured: 4.9e-4 rad over the first 50 steps, 7.3e-4 over the last 50 of 300)
    assert drift.max() < 5e-3, report
    assert drift[-50:].max() < drift[:50].max() + 2e-3, report
