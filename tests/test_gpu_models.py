"""Full-size networks on seeded synthetic weights against fixtures captured from the reference and
against the CPU oracle: waveform RMS error <= 1e-3 (north-star bar), f0 argmax exact outside near-ties."""
import os

import numpy as np
import pytest
import torch

import alive_oracle as O
from module import schema, synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda"
RMS_BAR = 1e-3


@pytest.fixture(scope="module")
def nets():
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    ce, pe, dec = ContentEncoder(seed=2).to(DEV), F0Estimator(seed=2).to(DEV), Decoder(seed=2).to(DEV)
    cpu = (synthetic.make_state_dict(schema.content_encoder_schema(), 2, "ce."),
           synthetic.make_state_dict(schema.f0_estimator_schema(), 2, "pe."),
           synthetic.make_state_dict(schema.decoder_schema(), 2, "dec."))
    return ce, pe, dec, cpu


def rms(a, b):
    return (a.double().cpu() - b.double().cpu()).pow(2).mean().sqrt().item()


@pytest.mark.parametrize("T", [5, 8, 24, 450])
def test_networks_against_reference_fixtures(golden_dir, nets, T):
    ce, pe, dec, cpu = nets
    z = np.load(os.path.join(golden_dir, f"full_T{T}.npz"))
    wav = torch.from_numpy(z["wav"])
    spec = torch.from_numpy(z["spec"]) if "spec" in z.files else O.spectrogram(wav)
    feat = ce(spec.to(DEV)).cpu()
    ref_feat = torch.from_numpy(z["feat"])
    f = feat if ref_feat.shape[1] == 768 else feat[:, ::8, :]
    assert rms(f, ref_feat) < 1e-5 * max(1.0, ref_feat.pow(2).mean().sqrt().item()), rms(f, ref_feat)
    f0 = pe.estimate(spec.to(DEV)).cpu()
    safe = torch.from_numpy(z["f0_margin"])[0] > 1e-4
    assert torch.equal(f0[0, 0][safe], torch.from_numpy(z["f0"])[0, 0][safe])
    assert safe.float().mean() > 0.9
    full_feat = O.content_encoder(cpu[0], spec)
    wave, phi = dec(full_feat.to(DEV), torch.from_numpy(z["f0_dec"]).to(DEV))
    err = rms(wave, torch.from_numpy(z["wave"]))
    assert err < RMS_BAR, f"waveform RMS error {err:.3e} vs reference fixture"
    # the carried phase is asin(sin(theta)): ill-conditioned where |sin| -> 1 (d asin = d sin / cos), so it is compared away
    # from the fold, at the tolerance the long-horizon streaming test measures (7e-4 rad after 300 carried steps)
    ref_phi = torch.from_numpy(z["phi_last"])
    ok = ref_phi.abs() < 1.45
    assert ok.float().mean() > 0.8
    torch.testing.assert_close(phi[:, :, -1].cpu()[ok], ref_phi[ok], rtol=0, atol=5e-3)
    torch.testing.assert_close(torch.sin(phi[:, :, -1].cpu()), torch.sin(ref_phi), rtol=0, atol=2e-3)      # everywhere: sin is what is carried


@pytest.fixture(scope="module")
def nets_of():
    """the three networks loaded with fixture weight set `tag` (synthetic.FIXTURE_SETS: two more seeds, one set with 4 x FiLM / pointwise
    weights), built once per set"""
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    cache = {}

    def get(tag):
        if tag not in cache:
            sds = synthetic.fixture_state_dicts(tag)
            mods = []
            for cls, sd in zip((ContentEncoder, F0Estimator, Decoder), sds):
                m = cls()
                m.load_state_dict(sd)
                mods.append(m.to(DEV))
            cache[tag] = (*mods, sds)
        return cache[tag]
    return get


@pytest.mark.parametrize("tag", sorted(synthetic.FIXTURE_SETS))
def test_networks_against_reference_fixtures_of_other_weight_sets(golden_dir, nets_of, tag):
    """Round 6: the 450-frame reference fixture on weight seeds 3 and 5 and on a set whose FiLM projections and
    pointwise convs carry 4 x the weights of a fresh initialisation (the Filter has no normalisation layer,
    /root/reference/module/decoder.py:153-195, and FiLM gains multiply activations, :112-117), on a voiced-speech-like input (harmonic
    stack + noise) with its smooth f0 contour.  Encoders in their default mode; the decoder (a) as a user gets it -- the precision mode
    CALIBRATED for the checkpoint (module/decoder.py: fp16 forms only where they stay within 1.5e-4 of split bf16 on a probe) --, (b) with
    the fp16 forms forced and (c) on split bf16.  Bars, relative to max(1, waveform RMS) (the x4 set's waveform has an RMS of 9.5):
    (a) 1e-4, (c) 3e-5, (b) 1e-4 on the two seeds and the path's 1e-3 on the x4 set, where every fp16 group is 15 - 25 x more sensitive
    (measured 8.6e-4: profiles/r06_decoder_precision_groups.json) -- which is exactly what the calibration has to catch.  Nothing may saturate."""
    from module import ops
    ce, pe, dec, cpu = nets_of(tag)
    if os.environ.get("ALIVE_DECODER_PRECISION") is not None:
        pytest.skip("the calibrated default is what this test is about")
    z = np.load(os.path.join(golden_dir, f"full_T450_{tag}.npz"))
    wav = torch.from_numpy(z["wav"])
    spec = O.spectrogram(wav)
    ops.f16_saturations(reset=True)
    feat = ce(spec.to(DEV)).cpu()
    ref_feat = torch.from_numpy(z["feat"])
    assert rms(feat[:, ::8, :], ref_feat) < 1e-5 * max(1.0, ref_feat.pow(2).mean().sqrt().item()), rms(feat[:, ::8, :], ref_feat)
    f0 = pe.estimate(spec.to(DEV)).cpu()
    safe = torch.from_numpy(z["f0_margin"])[0] > 1e-4
    assert torch.equal(f0[0, 0][safe], torch.from_numpy(z["f0"])[0, 0][safe])
    assert safe.float().mean() > 0.9
    full_feat = O.content_encoder(cpu[0], spec).to(DEV)
    f0d = torch.from_numpy(z["f0_dec"]).to(DEV)
    ref_wave = torch.from_numpy(z["wave"])
    scale = max(1.0, ref_wave.pow(2).mean().sqrt().item())
    errs = {}
    assert ops.decoder_precision(0) == 1 and dec.precision is None
    try:
        for what, prec in (("calibrated", None), ("fp16", 1), ("split", 2)):
            dec.precision = prec
            wave, phi = dec(full_feat, f0d)
            errs[what] = rms(wave, ref_wave)
            ref_phi = torch.from_numpy(z["phi_last"])
            torch.testing.assert_close(torch.sin(phi[:, :, -1].cpu()), torch.sin(ref_phi), rtol=0, atol=2e-3)
    finally:
        dec.precision = None
    cal = dec.calibration
    print(f"fixture set {tag}: waveform RMS {ref_wave.pow(2).mean().sqrt().item():.3f}, error calibrated {errs['calibrated']:.3e} "
          f"(probe: fp16 - split = {cal['relative_difference']:.2e} of the probe RMS -> mode {cal['chosen']}), fp16 forced {errs['fp16']:.3e}, "
          f"split bf16 {errs['split']:.3e}")
    assert cal["chosen"] == (2 if tag == "x4" else 1), cal
    assert errs["calibrated"] == (errs["split"] if cal["chosen"] == 2 else errs["fp16"])
    assert errs["calibrated"] < 1e-4 * scale and errs["split"] < 3e-5 * scale, errs
    assert errs["fp16"] < (RMS_BAR if tag == "x4" else 1e-4) * scale, errs
    assert ops.f16_saturations() == 0


def test_decoder_precision_calibration_is_per_checkpoint(nets, nets_of):
    """the calibration belongs to the Decoder object: the x4 checkpoint runs split bf16 while the seed-2 checkpoint of the same process
    keeps the fp16 forms; ALIVE_DECODER_PRECISION / ops.decoder_precision(2) / Decoder.precision override it; a new state_dict
    calibrates again"""
    from module import ops
    from module.decoder import Decoder
    if os.environ.get("ALIVE_DECODER_PRECISION") is not None:
        pytest.skip("a mode chosen through the environment switches the calibration off")
    _, _, dec2, _ = nets
    _, _, decx, sds = nets_of("x4")
    x = synthetic.gaussian("cal.x", 5, (1, 768, 128)).to(DEV)
    f0 = torch.full((1, 1, 128), 170.0, device=DEV)
    a = dec2(x, f0)[0]
    b = decx(x, f0)[0]
    assert dec2.calibration["chosen"] == 1 and decx.calibration["chosen"] == 2
    assert dec2.calibration["relative_difference"] < 1.0e-4 < 3.0e-4 < decx.calibration["relative_difference"]
    assert ops.decoder_precision(0) == 1                              # the process mode is untouched
    try:
        ops.decoder_precision(2)
        assert torch.equal(decx(x, f0)[0], b) and not torch.equal(dec2(x, f0)[0], a)
    finally:
        ops.decoder_precision(1)
    decx.precision = 1
    try:
        assert not torch.equal(decx(x, f0)[0], b)
    finally:
        decx.precision = None
    d = Decoder()
    d.load_state_dict(sds[2])
    d = d.to(DEV)
    assert d.calibration is None
    d(x, f0)
    assert d.calibration["chosen"] == 2
    d.load_state_dict(dec2.state_dict())
    assert d.calibration is None
    assert torch.equal(d(x, f0)[0], a) and d.calibration["chosen"] == 1


def test_decoder_stage_errors_are_small(nets):
    """same decoder inputs on both sides, batch of 3 windows of 40 frames: tight bound."""
    ce, pe, dec, cpu = nets
    x = synthetic.gaussian("dx", 4, (3, 768, 40))
    f0 = (100 + 300 * torch.from_numpy(synthetic.uniform01("df0", 4, 120)).float()).view(3, 1, 40)
    f0[1, 0, 10:14] = 0
    wave, _ = dec(x.to(DEV), f0.to(DEV))
    ref, _ = O.decoder(cpu[2], x, f0)
    assert rms(wave, ref) < 2e-4, rms(wave, ref)


def test_decoder_precision_modes(golden_dir, nets):
    """Round 5: the k = 5 convs of the 256- and 64-channel FilterBlocks (decoder.py:128-134), the pointwise convs of the feature
    extractor's ConvNeXt layers (common.py:74-82) and four smaller layers run on plain fp16 operands -- one MFMA per product --
    by default (alive_decoder_precision 1), on two-plane split bf16 in mode 2 (rounds 1 - 4).  Both against the oracle on the same
    decoder inputs, and against the reference's fixture of 450 frames: mode 1 stays inside 4e-5 (measured 2.9e-5), mode 2 inside 2e-5
    (5.0e-6); the bar of the path is 1e-3; the
    difference between the modes is the measured price of the plain form."""
    from module import ops
    ce, pe, dec, cpu = nets
    x = synthetic.gaussian("dx", 4, (3, 768, 40))
    f0 = (100 + 300 * torch.from_numpy(synthetic.uniform01("df0", 4, 120)).float()).view(3, 1, 40)
    f0[1, 0, 10:14] = 0
    ref, _ = O.decoder(cpu[2], x, f0)
    z = np.load(os.path.join(golden_dir, "full_T450.npz"))
    spec = torch.from_numpy(z["spec"]) if "spec" in z.files else O.spectrogram(torch.from_numpy(z["wav"]))
    feat450 = O.content_encoder(cpu[0], spec).to(DEV)
    f0_450 = torch.from_numpy(z["f0_dec"]).to(DEV)
    if os.environ.get("ALIVE_DECODER_PRECISION") is None:
        assert ops.decoder_precision(0) == 1
    out = {}
    try:
        for mode in (1, 2):
            assert ops.decoder_precision(mode) == mode
            out[mode] = (dec(x.to(DEV), f0.to(DEV))[0].cpu(), dec(feat450, f0_450)[0].cpu())
    finally:
        ops.decoder_precision(1 if os.environ.get("ALIVE_DECODER_PRECISION") != "2" else 2)
    e1, e2, d12 = rms(out[1][0], ref), rms(out[2][0], ref), rms(out[1][0], out[2][0])
    f1, f2 = rms(out[1][1], torch.from_numpy(z["wave"])), rms(out[2][1], torch.from_numpy(z["wave"]))
    print(f"decoder precision modes: 40 frames vs oracle plain {e1:.3e} split {e2:.3e} (plain - split {d12:.3e}); "
          f"450-frame fixture plain {f1:.3e} split {f2:.3e} (plain - split {rms(out[1][1], out[2][1]):.3e})")
    assert e2 < 2e-5 and e1 < 2e-5 and d12 < 2e-5, (e1, e2, d12)
    assert f1 < 4e-5 and f2 < 2e-5, (f1, f2)
    assert d12 > 1e-7                                  # the two modes really are different kernels


@pytest.mark.parametrize("switch", ["ALIVE_FB256=0", "ALIVE_FB64S=0", "ALIVE_FB256_WAVES=4"])
def test_decoder_forms_behind_the_filter_switches_stay_within_the_fixture_bar(golden_dir, switch):
    """Round 6: the 256- and 64-channel FilterBlocks of decoder precision mode 1 run on csrc/filter_big.hip; the kernels they replaced
    (conv by conv; filter_mid.hip's sweep kernel) and the one-wave-per-SIMD form stay behind environment switches that are read once per
    process -- one subprocess per switch decodes the reference's 450-frame fixture: same bar as the default form (4e-5 of max(1, RMS);
    measured 2.9e-5 each), no fp16 saturation, and the three forms really differ (or, for the wave count, do not: the sums are the same)."""
    import subprocess
    import sys
    child = r"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.join(%(root)r, "alive-vc_amd")); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import alive_oracle as O
from module import schema, synthetic, ops
from module.decoder import Decoder
z = np.load(os.path.join(%(gold)r, "full_T450.npz"))
sce = synthetic.make_state_dict(schema.content_encoder_schema(), 2, "ce.")
sdec = synthetic.make_state_dict(schema.decoder_schema(), 2, "dec.")
dec = Decoder(); dec.load_state_dict(sdec); dec = dec.to("cuda"); dec.precision = 1
feat = O.content_encoder(sce, O.spectrogram(torch.from_numpy(z["wav"]))).cuda()
wave, _ = dec(feat, torch.from_numpy(z["f0_dec"]).cuda())
ref = torch.from_numpy(z["wave"]).double()
err = (wave.double().cpu() - ref).pow(2).mean().sqrt().item() / max(1.0, ref.pow(2).mean().sqrt().item())
import hashlib
print("RESULT " + json.dumps({"err": err, "sat": ops.f16_saturations(reset=True), "digest": hashlib.sha256(wave.cpu().numpy().tobytes()).hexdigest()}))
"""
    root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    res = {}
    for name, env in (("default", {}), (switch, dict([switch.split("=")]))):
        clean = {k: v for k, v in os.environ.items() if k not in ("ALIVE_FB256", "ALIVE_FB64S", "ALIVE_FB256_WAVES")}
        r = subprocess.run([sys.executable, "-c", child % {"root": root, "gold": golden_dir}], env=dict(clean, **env), capture_output=True, text=True,
                           timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        assert line, (r.stdout[-1500:], r.stderr[-1500:])
        res[name] = __import__("json").loads(line[0][7:])
        assert res[name]["err"] < 4e-5 and res[name]["sat"] == 0, (name, res[name])
    same = res["default"]["digest"] == res[switch]["digest"]
    assert same == (switch == "ALIVE_FB256_WAVES=4"), (switch, res)


@pytest.mark.parametrize("n,lf", [(2, 27), (1, 15), (3, 33)])
def test_decoder_odd_frame_counts_in_both_precision_modes(nets, n, lf):
    """Odd frame counts: the 256-channel scale then has L = 10 Lf columns with L % 4 == 2, so its convs run WITHOUT plane operands -- the
    plain-fp16 kernel stages fp32 inputs itself (conv_split_kernel<128, 1, false, true>) and emits an fp32 second output; (1, 15) also
    keeps the frame-rate GEMMs on the fp32-activation path (15 columns) while the filter runs the batch kernels."""
    from module import ops
    ce, pe, dec, cpu = nets
    x = synthetic.gaussian(f"dxo{n}{lf}", 4, (n, 768, lf))
    f0 = (100 + 300 * torch.from_numpy(synthetic.uniform01(f"dfo{n}{lf}", 4, n * lf)).float()).view(n, 1, lf)
    ref, _ = O.decoder(cpu[2], x, f0)
    try:
        for mode, bar in ((1, 3e-5), (2, 2e-5)):
            ops.decoder_precision(mode)
            wave, _ = dec(x.to(DEV), f0.to(DEV))
            assert rms(wave, ref) < bar, (mode, rms(wave, ref))
    finally:
        ops.decoder_precision(1 if os.environ.get("ALIVE_DECODER_PRECISION") != "2" else 2)
    assert ops.f16_saturations() == 0


def test_realtime_two_steps(golden_dir, nets):
    """realtime_inference.py:146-167: phase carried through phi[:, :, end_of_output]."""
    from module.common import match_features
    from module.spectrogram import spectrogram
    from module import ops
    ce, pe, dec, cpu = nets
    z = np.load(os.path.join(golden_dir, "realtime_two_steps.npz"))
    stream = torch.from_numpy(z["stream"]).to(DEV)
    lib = synthetic.make_library(int(z["lib_M"]), int(z["lib_seed"])).to(DEV)
    c, bs, begin, end = int(z["chunk"]), int(z["buffersize"]), int(z["begin"]), int(z["end"])
    phi = 0
    for step in range(2):
        ring = stream[:, step * c: step * c + bs * c].contiguous()
        spec = spectrogram(ring)
        content = ce(spec)
        f0 = pe.estimate(spec)
        f0 = ops.pitch_transform_(f0, 1, f0_rate=float(z["f0_rate"]), pitch_shift=0.0)
        content = match_features(content, lib, k=4, alpha=0.0)
        data, phi_out = dec(content, f0=f0, phi=phi, crop=(begin, end))
        phi = phi_out[:, :, end].unsqueeze(2)
        assert rms(data, torch.from_numpy(z[f"wave{step}"])) < RMS_BAR
        ref_phi = torch.from_numpy(z[f"phi{step}"])
        ok = ref_phi.abs() < 1.45                                # away from the fold of asin (see the fixture test above)
        torch.testing.assert_close(phi.cpu()[ok], ref_phi[ok], rtol=0, atol=5e-3)
        torch.testing.assert_close(torch.sin(phi.cpu()), torch.sin(ref_phi), rtol=0, atol=2e-3)


def test_decoder_rejects_short_input_and_bad_scale(nets):
    ce, pe, dec, cpu = nets
    x = torch.zeros(1, 768, 4, device=DEV)
    with pytest.raises(ValueError):
        dec(x, torch.zeros(1, 1, 4, device=DEV))
    with pytest.raises(ValueError):
        dec(torch.zeros(1, 768, 8, device=DEV), torch.zeros(1, 1, 8, device=DEV), harmonics_scale=0.5)


def test_checkpoint_roundtrip(tmp_path, nets):
    from module.decoder import Decoder
    ce, pe, dec, cpu = nets
    p = tmp_path / "decoder.pt"
    torch.save(dec.state_dict(), p)
    sd = torch.load(p, map_location="cpu")
    assert list(sd.keys()) == list(schema.decoder_schema().keys())
    d2 = Decoder().to(DEV)
    d2.load_state_dict(sd)
    x = synthetic.gaussian("rt", 1, (1, 768, 6)).to(DEV)
    f0 = torch.full((1, 1, 6), 200.0, device=DEV)
    assert torch.equal(d2(x, f0)[0], dec(x, f0)[0])


def test_utterance_end_to_end(golden_dir, nets):
    """windowing + per-window pitch transform + kNN + decoder vs the reference's loop (inference.py:94-135)."""
    from module.pipeline import Converter, make_windows
    ce, pe, dec, cpu = nets
    z = np.load(os.path.join(golden_dir, "utterance_small.npz"))
    conv = Converter(ce, pe, dec).set_library(synthetic.make_library(int(z["lib_M"]), int(z["lib_seed"])))
    wf = torch.from_numpy(z["wf"])
    w_gpu, total = make_windows(wf.to(DEV), int(z["chunk"]))
    w_cpu, total_cpu = O.make_windows(wf, int(z["chunk"]))
    assert total == total_cpu and torch.equal(w_gpu.cpu(), w_cpu)
    out = conv.convert(wf, chunk=int(z["chunk"]), k=int(z["k"]), alpha=float(z["alpha"]), pitch_shift=float(z["pitch"]),
                       intonation=float(z["intonation"]), f0_rate=float(z["f0_rate"]))
    err = rms(out, torch.from_numpy(z["out"]))
    assert err < RMS_BAR, err


def test_default_window_end_to_end(nets):
    """one default 144000-sample window (450 frames), default CLI parameters, 5 k-vector library:
    the full device path against the CPU oracle on the same input."""
    from module.pipeline import Converter
    ce, pe, dec, cpu = nets
    lib = synthetic.make_library(5000, 21)
    wav = synthetic.make_waveform(144000, 77)
    conv = Converter(ce, pe, dec).set_library(lib)
    out = conv.convert_windows(wav.to(DEV), f0_rate=0.5)
    ref = O.convert_window(cpu[0], cpu[1], cpu[2], wav, lib, f0_rate=0.5)
    full, mid = rms(out, ref), rms(out[:, 48000:96000], ref[:, 48000:96000])
    print(f"450-frame window: rms error full {full:.3e}, kept centre third {mid:.3e}, signal rms {ref.pow(2).mean().sqrt():.3f}")
    assert mid < RMS_BAR, (full, mid)


def test_batch_invariance_of_the_networks(nets):
    """throughput mode runs 128 windows per launch: a window's result must not depend on its batch (bitwise)"""
    ce, pe, dec, _ = nets
    spec = torch.from_numpy(np.abs(synthetic.gaussian("bi.spec", 31, (3, 641, 450)).numpy())).to(DEV)
    feats, f0 = ce(spec), pe.estimate(spec)
    for i in range(3):
        assert torch.equal(ce(spec[i:i + 1].contiguous()), feats[i:i + 1])
        assert torch.equal(pe.estimate(spec[i:i + 1].contiguous()), f0[i:i + 1])
    f0 = 100.0 + 20.0 * f0 / 4096.0
    wave, _ = dec(feats, f0=f0)
    for i in (0, 2):
        w1, _ = dec(feats[i:i + 1].contiguous(), f0=f0[i:i + 1].contiguous())
        assert torch.equal(w1, wave[i:i + 1])


def test_context_trim_keeps_the_centre_third_bitwise(nets):
    """inference.py keeps the centre third of every window: matching only the frames that can reach it through the
    decoder must not change a single kept sample"""
    from module.pipeline import Converter, make_windows, stitch
    ce, pe, dec, _ = nets
    conv = Converter(ce, pe, dec, DEV).set_library(synthetic.make_library(3000, 9).to(DEV))
    wf = (0.3 * synthetic.make_waveform(16000 * 7, 41)).to(DEV)
    for chunk in (48000, 16000, 4800):           # 150 / 50 / 15 frames per chunk (the last: nothing to trim)
        full = conv.convert(wf, chunk=chunk, k=4, alpha=0.1)
        trimmed = conv.convert(wf, chunk=chunk, k=4, alpha=0.1, trim_context=True)
        assert torch.equal(full, trimmed), chunk
    windows, _ = make_windows(wf, 48000)
    a = conv.convert_windows(windows, k=4)
    b = conv.convert_windows(windows, k=4, keep_frames=(150, 300))
    assert torch.equal(a[:, 48000:96000], b[:, 48000:96000]) and not torch.equal(a, b)     # the context thirds do differ


def test_decoder_forward_range_is_bitwise_inside_its_margins(nets):
    """alive_decoder_forward_range: the decoder on frames [118, 316) of 450-frame windows reproduces frames [150, 300) of the
    whole-window decode bit for bit (interpolation coordinates and oscillator phase are those of the whole window)"""
    _, _, dec, _ = nets
    x = synthetic.gaussian("dr.x", 51, (2, 768, 450)).to(DEV)
    f0 = (90.0 + 60.0 * torch.from_numpy(synthetic.uniform01("dr.f0", 52, 2 * 450)).float()).view(2, 1, 450).to(DEV)
    full, _ = dec(x, f0=f0)
    a, b = 118, 316
    part = dec.forward_range(x[:, :, a:b].contiguous(), f0, a)
    assert part.shape == (2, (b - a) * 320)
    assert torch.equal(part[:, (150 - a) * 320:(300 - a) * 320], full[:, 150 * 320:300 * 320])
    # a range that starts at the window's first frame keeps the left edge exact as well
    head = dec.forward_range(x[:, :, :200].contiguous(), f0, 0)
    assert torch.equal(head[:, :180 * 320], full[:, :180 * 320])
    with pytest.raises(ValueError):
        dec.forward_range(x[:, :, :10].contiguous(), f0, 445)


@pytest.mark.parametrize("n,t", [(3, 450), (1, 8), (2, 97)])
def test_network_entry_points_stay_inside_their_workspaces(nets, n, t):
    """alive_spectrogram / alive_content_encoder / alive_f0_estimate / alive_decoder_forward on caller-owned memory with guard
    bands behind every workspace and output: nothing outside `*_workspace_bytes` and the documented output shapes is written
    (batch kernels at 3 x 450 and 2 x 97 frames, streaming kernels at 8 frames)"""
    from module import _native as nat
    from module import spectrogram as sp
    ce, pe, dec, _ = nets
    L = nat.lib()
    st = nat.stream()
    G = 1 << 20

    def banded(nbytes, dtype=torch.uint8, fill=0xAB):
        return torch.full((int(nbytes) + G,), fill, dtype=dtype, device=DEV)

    def intact(buf, nbytes, fill=0xAB):
        return bool((buf[int(nbytes):] == fill).all())

    wav = (0.3 * synthetic.make_waveform(t * 320, 77)).repeat(n, 1).to(DEV).contiguous()
    sp.spectrogram(wav[:1])                                             # builds the cached DFT basis
    basis = sp._basis[str(wav.device)]
    need = L.alive_spectrogram_workspace_bytes(n, t * 320)
    ws = banded(need)
    spec = torch.full((n * 641 * t + 4096,), 7.0, device=DEV)
    nat.check(L.alive_spectrogram(nat.ptr(basis), nat.ptr(wav), n, t * 320, spec.data_ptr(), ws.data_ptr(), st), "spectrogram")
    torch.cuda.synchronize()
    assert intact(ws, need) and bool((spec[n * 641 * t:] == 7.0).all())
    spec = spec[:n * 641 * t].view(n, 641, t).contiguous()
    assert torch.equal(spec, sp.spectrogram(wav))

    for net, fn, wsq, ch in ((ce, L.alive_content_encoder, L.alive_content_encoder_workspace_bytes, 768),
                             (pe, L.alive_f0_estimate, L.alive_f0_estimate_workspace_bytes, 1)):
        need = wsq(n, t)
        ws = banded(need)
        out = torch.full((n * ch * t + 4096,), 7.0, device=DEV)
        nat.check(fn(net.table().array, nat.ptr(spec), n, t, out.data_ptr(), ws.data_ptr(), st), "net")
        torch.cuda.synchronize()
        assert intact(ws, need), "network wrote behind its workspace"
        assert bool((out[n * ch * t:] == 7.0).all())

    need = L.alive_decoder_workspace_bytes(n, t)
    ws = banded(need)
    x = synthetic.gaussian("gb.x", 4, (n, 768, t)).to(DEV)
    f0 = torch.full((n, 1, t), 150.0, device=DEV)
    wave = torch.full((n * t * 320 + 4096,), 7.0, device=DEV)
    phi = torch.full((n * 64 + 4096,), 7.0, device=DEV)
    nat.check(L.alive_decoder_forward(dec.table().array, nat.ptr(x), nat.ptr(f0), None, 0, t * 320 - 1, n, t, wave.data_ptr(),
                                      phi.data_ptr(), ws.data_ptr(), st), "decoder")
    torch.cuda.synchronize()
    assert intact(ws, need), "decoder wrote behind its workspace"
    assert bool((wave[n * t * 320:] == 7.0).all() and (phi[n * 64:] == 7.0).all())
    assert torch.isfinite(wave[:n * t * 320]).all()


def test_f0_estimator_forward_returns_the_reference_logits(nets):
    """F0Estimator.forward (/root/reference/module/f0_estimator.py:22-27): class logits [N, 4096, T], layer by layer through the
    op-level C ABI; their argmax is what the fused `estimate` returns"""
    _, pe, _, cpu = nets
    spec = O.spectrogram(synthetic.make_waveform(320 * 24, 50 + 24))
    lg = pe(spec.to(DEV))
    ref = O.f0_logits(cpu[1], spec)
    assert lg.shape == ref.shape == (1, 4096, 24)
    assert rms(lg, ref) < 1e-5 * max(1.0, ref.pow(2).mean().sqrt().item())
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(lg.argmax(1).cpu()[safe], ref.argmax(1)[safe])
    assert torch.equal(pe.estimate(spec.to(DEV))[:, 0].cpu()[safe], ref.argmax(1).float()[safe])


@pytest.mark.parametrize("sizes", [(1280, 64, 96, 48, 2), (1280, 96, 160, 200, 3)])
def test_encoders_with_non_default_sizes_run_the_generic_path(sizes):
    """the reference's constructors take sizes (content_encoder.py:9-14, f0_estimator.py:9-14): anything but the defaults runs layer
    by layer through alive_conv1d / alive_dwconv_norm / alive_channel_norm and matches the oracle on the same state_dict"""
    from module.content_encoder import ContentEncoder
    from module.f0_estimator import F0Estimator
    n_fft, c, hdim, out, layers = sizes
    spec = O.spectrogram(synthetic.make_waveform(320 * 40, 7))
    ce = ContentEncoder(n_fft, c, hdim, out, layers, seed=5).to(DEV)
    assert ce.generic and set(ce.state_dict()) == set(schema.content_encoder_schema(c, hdim, out, layers))
    sd = {k: v.cpu() for k, v in ce.state_dict().items()}
    ref = O.content_encoder(sd, spec)
    got = ce(spec.to(DEV))
    assert got.shape == ref.shape == (1, out, 40)
    assert rms(got, ref) < 2e-5 * max(1.0, ref.pow(2).mean().sqrt().item()), rms(got, ref)
    pe = F0Estimator(n_fft, c, hdim, out, layers, seed=6).to(DEV)
    sd = {k: v.cpu() for k, v in pe.state_dict().items()}
    ref = O.f0_logits(sd, spec)
    lg = pe(spec.to(DEV))
    assert rms(lg, ref) < 2e-5 * max(1.0, ref.pow(2).mean().sqrt().item())
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(pe.estimate(spec.to(DEV))[:, 0].cpu()[safe], ref.argmax(1).float()[safe])
    with pytest.raises(RuntimeError):
        ContentEncoder(n_fft, c, hdim, out, layers, seed=5)(spec)          # no CPU path


@pytest.mark.parametrize("n,t", [(1, 150), (3, 97), (2, 450), (130, 450)])
def test_fused_front_end_equals_the_three_separate_calls_bitwise(nets, n, t):
    """SURVEY 8 f1 (round 5): alive_front_end -- the DFT GEMM whose epilogue leaves the magnitudes as the plane-packed operand of
    the input layers (no fp32 spectrogram), the ContentEncoder and F0Estimator input layers as ONE GEMM with two outputs -- against
    spectrogram() -> ce() / pe.estimate() (/root/reference/module/spectrogram.py:5-10, content_encoder.py:21-25,
    f0_estimator.py:22-34): same bits, inside its workspace, for ragged column counts (291 columns = 2 tiles + 35) and a batch
    whose row / column tiles exceed one round of the chip."""
    from module import _native as nat
    from module import ops
    from module import spectrogram as sp
    ce, pe, _, _ = nets
    wav = torch.stack([0.3 * synthetic.make_waveform(t * 320, 500 + i % 7)[0] * (1.0 + 0.01 * i) for i in range(n)]).to(DEV).contiguous()
    spec = sp.spectrogram(wav)
    feat_ref, f0_ref = ce(spec), pe.estimate(spec)
    feat, f0 = ops.front_end(wav, ce, pe)
    assert torch.equal(feat, feat_ref) and torch.equal(f0, f0_ref)
    # caller-owned memory with guard bands, through the C ABI
    L = nat.lib()
    need = L.alive_front_end_workspace_bytes(n, t * 320)
    G = 1 << 20
    ws = torch.full((need + G,), 0xAB, dtype=torch.uint8, device=DEV)
    o1 = torch.full((n * 768 * t + 4096,), 7.0, device=DEV)
    o2 = torch.full((n * t + 4096,), 7.0, device=DEV)
    w, b = ops._front[(id(ce.table()), id(pe.table()))][:2]
    nat.check(L.alive_front_end(nat.ptr(sp.dft_basis(wav.device)), ce.table().array, pe.table().array, nat.ptr(w), nat.ptr(b), nat.ptr(wav),
                                n, t * 320, o1.data_ptr(), o2.data_ptr(), ws.data_ptr(), nat.stream()), "alive_front_end")
    torch.cuda.synchronize()
    assert bool((ws[need:] == 0xAB).all()), "the fused front end wrote behind its workspace"
    assert bool((o1[n * 768 * t:] == 7.0).all() and (o2[n * t:] == 7.0).all())
    assert torch.equal(o1[:n * 768 * t].view(n, 768, t), feat_ref) and torch.equal(o2[:n * t].view(n, 1, t), f0_ref)


def test_fused_front_end_falls_back_for_a_handful_of_frames(nets):
    """fewer than 96 frame columns (the streaming ring) or a length that is not a multiple of 8: the three separate calls"""
    from module import ops
    from module import spectrogram as sp
    ce, pe, _, _ = nets
    wav = (0.3 * synthetic.make_waveform(8 * 320, 9)).to(DEV)
    spec = sp.spectrogram(wav)
    feat, f0 = ops.front_end(wav, ce, pe)
    assert torch.equal(feat, ce(spec)) and torch.equal(f0, pe.estimate(spec))


def test_a_checkpoint_outside_fp16_range_is_refused_not_mis_converted():
    """The encoders' fp16 split planes hold 256 x the normalised activations: a ChannelNorm gain beyond ~11 CAN drive them past 65504 (beyond ~60 it does on ordinary data).  Such a
    checkpoint is saturated AND counted (alive_f16_saturations); Converter.check_fp16_range raises, and encoder precision mode 2 (three bf16
    planes: fp32's range) converts it like the oracle does."""
    from module import ops
    from module.content_encoder import ContentEncoder
    from module.pipeline import Converter
    sd = synthetic.make_state_dict(schema.content_encoder_schema(), 2)
    sd["mid_layers.1.norm.scale"] = sd["mid_layers.1.norm.scale"] * 200.0         # normalised values reach ~4: |y| ~ 800 > 255 = 65504 / 2^8
    ce = ContentEncoder()
    ce.load_state_dict(sd)
    ce = ce.to(DEV)
    spec = O.spectrogram(synthetic.make_waveform(144000, 5))
    ref = O.content_encoder(sd, spec)
    ops.f16_saturations(reset=True)
    try:
        ops.encoder_precision(1)
        ce(spec.to(DEV))
        assert ops.f16_saturations() > 0
        with pytest.raises(RuntimeError, match="fp16"):
            Converter.check_fp16_range()
        ops.encoder_precision(2)
        feat = ce(spec.to(DEV)).cpu()
        assert ops.f16_saturations() == 0
        assert rms(feat, ref) < 1e-5 * max(1.0, ref.pow(2).mean().sqrt().item())
    finally:
        ops.encoder_precision(1 if os.environ.get("ALIVE_ENCODER_PRECISION") != "2" else 2)
        ops.f16_saturations(reset=True)


def test_a_saturating_batch_is_repeated_on_bf16_planes(nets):
    """Round 6: every public batch entry point runs under ops.Fp16Guard -- counters cleared in stream order when the batch
    starts (stale counts of earlier direct calls do not matter), read after a device synchronisation when it ends, and a batch that left
    fp16's range is REPEATED in precision modes 2 instead of being returned saturated: the result is bitwise that of a process started
    with ALIVE_ENCODER_PRECISION=2 ALIVE_DECODER_PRECISION=2, the modes in force before are restored."""
    import warnings
    from module import ops
    from module.content_encoder import ContentEncoder
    from module.pipeline import Converter
    _, pe, dec, _ = nets
    sd = synthetic.make_state_dict(schema.content_encoder_schema(), 2, "ce.")
    sd["mid_layers.1.norm.scale"] = sd["mid_layers.1.norm.scale"] * 200.0
    ce = ContentEncoder()
    ce.load_state_dict(sd)
    conv = Converter(ce.to(DEV), pe, dec).set_library(synthetic.make_library(2000, 3))
    wav = (0.3 * synthetic.make_waveform(144000, 5)).to(DEV)
    enc0, dec0 = ops.encoder_precision(0), ops.decoder_precision(0)
    if enc0 == 2:
        pytest.skip("the encoder already runs on bf16 planes")
    x = synthetic.gaussian("stale", 1, (1, 64, 130))
    x[0, 3, 7] = 1e9
    ops.to_planes(x.to(DEV), 1)                                   # a stale saturation from a direct low-level call
    before = ops.Fp16Guard.fallbacks
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        out = conv.convert_windows(wav, k=4)
    assert ops.Fp16Guard.fallbacks == before + 1 and any("fp16" in str(w.message) for w in rec)
    assert (ops.encoder_precision(0), ops.decoder_precision(0)) == (enc0, dec0)
    assert ops.f16_saturations() == 0
    try:
        ops.encoder_precision(2)
        ops.decoder_precision(2)
        want = conv.convert_windows(wav, k=4)
    finally:
        ops.encoder_precision(enc0)
        ops.decoder_precision(dec0)
    assert torch.equal(out, want)
    # and a clean batch right after a stale count is NOT repeated
    ops.to_planes(x.to(DEV), 1)
    conv2 = Converter(nets[0], pe, dec).set_library(conv.library)
    before = ops.Fp16Guard.fallbacks
    conv2.convert_windows(wav, k=4)
    assert ops.Fp16Guard.fallbacks == before
