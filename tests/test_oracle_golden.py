"""The CPU oracle against fixtures captured from the reference itself
(oracle/gen_golden.py).  Runs on CPU; tolerances allow for a different host
CPU ISA than the one the fixtures were generated on."""
import json
import os

import numpy as np
import pytest
import torch

import alive_oracle as O
from module import schema, synthetic

TOL = dict(rtol=2e-5, atol=2e-5)


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    d = {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}
    sd = {"n." + k[3:]: v for k, v in d.items() if k.startswith("w::")}
    return d, sd


@pytest.fixture(scope="module")
def nets(golden_dir):
    ce = synthetic.make_state_dict(schema.content_encoder_schema(), 2, "ce.")
    pe = synthetic.make_state_dict(schema.f0_estimator_schema(), 2, "pe.")
    dec = synthetic.make_state_dict(schema.decoder_schema(), 2, "dec.")
    return ce, pe, dec


def test_schema_matches_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "reference_state_dict_schema.json")))
    for name, sch in (("content_encoder", schema.content_encoder_schema()),
                      ("f0_estimator", schema.f0_estimator_schema()),
                      ("decoder", schema.decoder_schema())):
        mine = {k: list(v[0]) for k, v in sch.items()}
        assert mine == ref[name], name
        assert list(mine.keys()) == list(ref[name].keys()), name + " key order"
    assert ref["voice_library"] == {"tokens": [1, 768, 512]}


def test_synthetic_weights_are_bit_reproducible(golden_dir, nets):
    want = json.load(open(os.path.join(golden_dir, "weights_sha256.json")))
    ce, pe, dec = nets
    assert synthetic.state_dict_digest(ce) == want["content_encoder"]
    assert synthetic.state_dict_digest(pe) == want["f0_estimator"]
    assert synthetic.state_dict_digest(dec) == want["decoder"]


@pytest.mark.parametrize("tag", list("abcdef"))
def test_knn(golden_dir, tag):
    d, _ = load(golden_dir, "knn_" + tag)
    T, M, k, alpha = d["T"], d["M"], d["k"], d["alpha"]
    if M <= 1000:
        src, lib = d["src"], synthetic.make_library(M, 12)
    else:
        src = synthetic.gaussian(f"knn.src.{tag}", 11, (1, 768, T))
        lib = synthetic.make_library(M, 12)
        if d["kind"] == "clustered":
            base = synthetic.gaussian("knn.base", 13, (1, 768, 1))
            lib = base + 0.35 * lib
            src = base + 0.35 * src
    out, idx, _ = O.match_features(src, lib, k, alpha, return_indices=True)
    safe = d["gap"] > 1e-5
    got = np.sort(idx[0].numpy(), axis=1)[safe.numpy()]
    want = np.sort(d["idx"].numpy(), axis=1)[safe.numpy()]
    assert safe.float().mean() > 0.9
    assert np.array_equal(got, want)
    ref_out = d["out"]
    o = out if M <= 1000 else out[:, ::16, :]
    torch.testing.assert_close(o[:, :, safe], ref_out[:, :, safe], rtol=1e-5, atol=1e-6)


def test_knn_m_smaller_than_k_raises():
    with pytest.raises(RuntimeError):
        O.match_features(synthetic.gaussian("x", 1, (1, 768, 3)), synthetic.make_library(3, 1), k=4)


def test_voice_library_match(golden_dir):
    d, _ = load(golden_dir, "voice_library_match")
    out = O.voice_library_match(synthetic.make_library(512, d["seed"]), d["src"], 4, 0.25)
    torch.testing.assert_close(out, d["out"], rtol=1e-5, atol=1e-6)


def test_blocks(golden_dir):
    d, sd = load(golden_dir, "blk_channel_norm")
    torch.testing.assert_close(O.channel_norm(sd, "n", d["x"]), d["y"], **TOL)
    d, sd = load(golden_dir, "blk_adaptive_channel_norm")
    torch.testing.assert_close(O.adaptive_channel_norm(sd, "n", d["x"], d["c"]), d["y"], **TOL)
    d, sd = load(golden_dir, "blk_convnext")
    torch.testing.assert_close(O.convnext1d(sd, "n", d["x"]), d["y"], **TOL)
    d, sd = load(golden_dir, "blk_adaptive_convnext")
    torch.testing.assert_close(O.convnext1d(sd, "n", d["x"], cond=d["c"]), d["y"], **TOL)
    for dil in (1, 2, 4):
        d, sd = load(golden_dir, f"blk_causal_conv_d{dil}")
        torch.testing.assert_close(O.causal_conv1d(sd, "n", d["x"], dil), d["y"], **TOL)
    d, sd = load(golden_dir, "blk_modulated_causal_conv")
    torch.testing.assert_close(O.modulated_causal_conv(sd, "n", d["x"], d["c"], 2), d["y"], **TOL)
    d, sd = load(golden_dir, "blk_filter_res_block")
    torch.testing.assert_close(O.filter_res_block(sd, "n", d["x"], d["c"], 4), d["y"], **TOL)
    d, sd = load(golden_dir, "blk_filter_block")
    torch.testing.assert_close(O.filter_block(sd, "n", d["x"], d["c"]), d["y"], **TOL)
    d, sd = load(golden_dir, "blk_f0_encoder")
    torch.testing.assert_close(O.f0_encoder(sd, "n", d["f0"]), d["y"], rtol=1e-3, atol=1e-3)
    d, sd = load(golden_dir, "blk_filter")
    torch.testing.assert_close(O.source_filter(sd, "n", d["src"], d["c"]), d["y"], **TOL)


def test_oscillator(golden_dir):
    d, sd = load(golden_dir, "blk_oscillator")
    w, ph = O.harmonic_oscillator(sd, "n", d["x"], d["f0"])
    torch.testing.assert_close(w, d["wave"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ph[:, :, 3000], d["phi_col"], rtol=1e-3, atol=1e-3)
    d, sd = load(golden_dir, "blk_oscillator_carry")
    w, ph = O.harmonic_oscillator(sd, "n", d["x"], d["f0"], phi=d["phi_in"], crop0=d["crop0"])
    torch.testing.assert_close(w, d["wave"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("T", [5, 8, 24])
def test_full_models(golden_dir, nets, T):
    ce, pe, dec = nets
    d, _ = load(golden_dir, f"full_T{T}")
    spec = O.spectrogram(d["wav"])
    torch.testing.assert_close(spec, d["spec"], rtol=1e-4, atol=1e-3)
    feat = O.content_encoder(ce, d["spec"])
    torch.testing.assert_close(feat, d["feat"], **TOL)
    f0 = O.f0_estimate(pe, d["spec"])
    safe = d["f0_margin"][0] > 1e-4
    assert torch.equal(f0[0, 0][safe], d["f0"][0, 0][safe])
    wave, phi = O.decoder(dec, d["feat"], d["f0_dec"])
    rms = (wave - d["wave"]).pow(2).mean().sqrt().item()
    assert rms < 1e-4, rms


def test_windowing(golden_dir):
    want = json.load(open(os.path.join(golden_dir, "windowing.json")))
    for L, (n, w, total, s) in want.items():
        win, tot = O.make_windows(synthetic.make_waveform(int(L), 60), 48000)
        assert (win.shape[0], win.shape[1], tot) == (n, w, total)
        assert abs(float(win.double().abs().sum()) - s) < 1e-6 * max(1.0, s)


def test_utterance_and_realtime(golden_dir, nets):
    ce, pe, dec = nets
    d, _ = load(golden_dir, "utterance_small")
    lib = synthetic.make_library(d["lib_M"], d["lib_seed"])
    out = O.convert_utterance(ce, pe, dec, d["wf"], lib, chunk=d["chunk"], k=d["k"], alpha=d["alpha"],
                              pitch_shift=d["pitch"], intonation=d["intonation"], f0_rate=d["f0_rate"])
    assert (out - d["out"]).pow(2).mean().sqrt().item() < 1e-4
    d, _ = load(golden_dir, "realtime_two_steps")
    lib = synthetic.make_library(d["lib_M"], d["lib_seed"])
    phi = 0
    for step in range(2):
        c, bs = d["chunk"], d["buffersize"]
        ring = d["stream"][:, step * c: step * c + bs * c]
        wave, phi = O.realtime_step(ce, pe, dec, ring, lib, phi, d["begin"], d["end"], f0_rate=d["f0_rate"])
        assert (wave - d[f"wave{step}"]).pow(2).mean().sqrt().item() < 1e-4
        torch.testing.assert_close(phi, d[f"phi{step}"], rtol=1e-3, atol=1e-3)
