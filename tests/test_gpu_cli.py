"""The drop-in CLI surface on the MI355X: inference.py / realtime_inference.py / generate_voice_library.py
with the reference's flags, checkpoint files and library format, checked against the CPU oracle."""
import os
import sys

import numpy as np
import pytest
import torch

import alive_oracle as O
from module import audio_io, schema, synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alive-vc_amd"))


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    sds = {"content_encoder.pt": synthetic.make_state_dict(schema.content_encoder_schema(), 2, "ce."),
           "f0_estimator.pt": synthetic.make_state_dict(schema.f0_estimator_schema(), 2, "pe."),
           "decoder.pt": synthetic.make_state_dict(schema.decoder_schema(), 2, "dec.")}
    for name, sd in sds.items():
        torch.save(sd, d / name)
    torch.save({"tokens": synthetic.make_library(512, 5)}, d / "voice_library.pt")     # reference M = 512 file
    os.makedirs(d / "inputs")
    wav24 = synthetic.make_waveform(24000, 91) * 0.5            # 1 s mono at 24 kHz (BASELINE config 1 shape)
    audio_io.save(str(d / "inputs" / "utt.wav"), wav24, 24000)
    return d, sds, wav24


def test_inference_cli_matches_oracle(workdir):
    import inference
    d, sds, wav24 = workdir
    inference.main(["-i", str(d / "inputs"), "-o", str(d / "outputs"), "-dep", str(d / "decoder.pt"),
                    "-cep", str(d / "content_encoder.pt"), "-f0ep", str(d / "f0_estimator.pt"),
                    "-lib", str(d / "voice_library.pt"), "-d", "cuda", "-c", "4800", "-f0", "0.5", "-p", "2", "-a", "0.1"])
    out, sr = audio_io.load(str(d / "outputs" / "0_utt.wav"))
    assert sr == 24000 and out.shape == (1, 24000)
    # oracle: torch formulation of the resampler / gain on CPU, reference loop in between
    wf = O.resample(wav24, 24000, 16000)
    wf = (wf / wf.abs().max()).mean(dim=0, keepdim=True)
    ref = O.convert_utterance(sds["content_encoder.pt"], sds["f0_estimator.pt"], sds["decoder.pt"], wf,
                              synthetic.make_library(512, 5), chunk=4800, k=4, alpha=0.1, pitch_shift=2.0, f0_rate=0.5)
    ref = O.gain(O.resample(ref, 16000, 24000), 1.0)
    err = (out - ref).pow(2).mean().sqrt().item()
    assert err < 1e-3, err


def test_inference_cli_baseline_config_1_default_chunk_and_1k_library(workdir):
    """BASELINE config 1 as written: a 1 s mono 24 kHz utterance, a 1 000-vector library, every flag at its default (chunk
    48 000 -> 3 overlapping windows of 450 frames = 1 350 computed frames for 50 useful ones; k = 4, alpha = 0)
    -- except the device, which this build only accepts as `cuda` (the reference's config 1 says `-d cpu`: there is no CPU
    path here by contract).  Output against the oracle's restatement of inference.py:86-142."""
    import inference
    d, sds, wav24 = workdir
    lib1k = synthetic.make_library(1000, 1)
    torch.save({"tokens": lib1k}, d / "voice_library_1k.pt")
    inference.main(["-i", str(d / "inputs"), "-o", str(d / "out_cfg1"), "-dep", str(d / "decoder.pt"),
                    "-cep", str(d / "content_encoder.pt"), "-f0ep", str(d / "f0_estimator.pt"),
                    "-lib", str(d / "voice_library_1k.pt"), "-d", "cuda"])
    out, sr = audio_io.load(str(d / "out_cfg1" / "0_utt.wav"))
    assert sr == 24000 and out.shape == (1, 24000)
    wf = O.resample(wav24, 24000, 16000)
    wf = (wf / wf.abs().max()).mean(dim=0, keepdim=True)
    windows, _ = O.make_windows(wf, 48000)
    assert windows.shape == (3, 144000)                                   # the 1 350 frames of SURVEY F10
    ref = O.convert_utterance(sds["content_encoder.pt"], sds["f0_estimator.pt"], sds["decoder.pt"], wf, lib1k, chunk=48000,
                              k=4, alpha=0.0)
    ref = O.gain(O.resample(ref, 16000, 24000), 1.0)
    err = (out - ref).pow(2).mean().sqrt().item()
    assert err < 1e-3, err


def test_inference_cli_target_utterance_stereo_input_gain_and_normalize(workdir):
    """inference.py:69-84,88-93,136-142: a target utterance (-t) whose content frames join the library in front of the -lib
    tokens, a stereo 22.05 kHz input (normalised over both channels, then averaged), output gain and -norm"""
    import inference
    d, sds, _ = workdir
    os.makedirs(d / "in_st", exist_ok=True)
    st = torch.cat([synthetic.make_waveform(22050, 92) * 0.4, synthetic.make_waveform(22050, 93) * 0.7], 0)      # [2, 1 s]
    audio_io.save(str(d / "in_st" / "pair.wav"), st, 22050)
    tgt_wav = synthetic.make_waveform(16000 * 2, 94) * 0.6                                                       # 2 s at 16 kHz
    audio_io.save(str(d / "target.wav"), tgt_wav, 16000)
    inference.main(["-i", str(d / "in_st"), "-o", str(d / "out_st"), "-dep", str(d / "decoder.pt"), "-cep", str(d / "content_encoder.pt"),
                    "-f0ep", str(d / "f0_estimator.pt"), "-lib", str(d / "voice_library.pt"), "-t", str(d / "target.wav"),
                    "-d", "cuda", "-c", "6400", "-g", "-3.0", "-norm", "True", "-k", "5"])
    out, sr = audio_io.load(str(d / "out_st" / "0_pair.wav"))
    assert sr == 22050 and out.shape == (1, 22050)
    ce, pe, dec = sds["content_encoder.pt"], sds["f0_estimator.pt"], sds["decoder.pt"]
    tw = tgt_wav / tgt_wav.abs().max()
    lib = torch.cat([O.content_encoder(ce, O.spectrogram(tw[:1])), synthetic.make_library(512, 5)], dim=2)
    wf = O.resample(st, 22050, 16000)
    wf = (wf / wf.abs().max()).mean(dim=0, keepdim=True)
    ref = O.convert_utterance(ce, pe, dec, wf, lib, chunk=6400, k=5, alpha=0.0)
    ref = O.gain(O.resample(ref, 16000, 22050), -3.0)
    peak = ref.abs().max().item()
    ref = ref / peak
    err = (out - ref).pow(2).mean().sqrt().item()
    assert err < 1e-3 / peak, (err, peak)        # -norm rescales the waveform to peak 1: the 1e-3 bar scales with it


def test_inference_cli_strict_knn_writes_the_same_file(workdir, monkeypatch):
    """--knn-strict (this build only): the kNN match under the deterministic certificate.  Same neighbours, same file."""
    import inference
    from module import common
    monkeypatch.setenv("ALIVE_KNN_STRICT", "0")              # main() sets the variable: restored when the test ends
    d, _, _ = workdir
    base = ["-i", str(d / "inputs"), "-dep", str(d / "decoder.pt"), "-cep", str(d / "content_encoder.pt"), "-f0ep", str(d / "f0_estimator.pt"),
            "-lib", str(d / "voice_library.pt"), "-d", "cuda", "-c", "6400", "-a", "0.05"]
    inference.main(base + ["-o", str(d / "out_default")])
    inference.main(base + ["-o", str(d / "out_strict"), "--knn-strict"])
    assert common._strict()
    a, _ = audio_io.load(str(d / "out_default" / "0_utt.wav"))
    b, _ = audio_io.load(str(d / "out_strict" / "0_utt.wav"))
    assert torch.equal(a, b)
    common.forget_packed()


def test_inference_cli_trim_context_writes_the_same_file(workdir):
    import inference
    d, _, _ = workdir
    base = ["-i", str(d / "inputs"), "-dep", str(d / "decoder.pt"), "-cep", str(d / "content_encoder.pt"), "-f0ep", str(d / "f0_estimator.pt"),
            "-lib", str(d / "voice_library.pt"), "-d", "cuda", "-c", "6400", "-p", "-1", "-a", "0.05"]      # 20 frames per chunk
    inference.main(base + ["-o", str(d / "out_full"), "--no-trim-context", "--no-share-overlap"])     # the reference's order of work
    inference.main(base + ["-o", str(d / "out_trim"), "--trim-context"])
    inference.main(base + ["-o", str(d / "out_default")])                                             # trimmed (+ shared when it pays)
    a, _ = audio_io.load(str(d / "out_full" / "0_utt.wav"))
    b, _ = audio_io.load(str(d / "out_trim" / "0_utt.wav"))
    c, _ = audio_io.load(str(d / "out_default" / "0_utt.wav"))
    assert torch.equal(a, b) and torch.equal(a, c)


def test_inference_cli_overlap_sharing_writes_the_same_file(workdir):
    """the CLI shares the front end between the overlapping windows of a long utterance (share_overlap="auto": a 1 s file
    stays on the per-window order, a 60 s file against a 1 M library is shared); --no-share-overlap (the reference's
    per-window order) must write the same samples either way"""
    import inference
    d, _, _ = workdir
    base = ["-i", str(d / "inputs"), "-dep", str(d / "decoder.pt"), "-cep", str(d / "content_encoder.pt"), "-f0ep", str(d / "f0_estimator.pt"),
            "-lib", str(d / "voice_library.pt"), "-d", "cuda", "-c", "16000", "-p", "1.5", "-a", "0.05"]      # 50 frames per chunk
    inference.main(base + ["-o", str(d / "out_shared")])
    inference.main(base + ["-o", str(d / "out_per_window"), "--no-share-overlap", "--no-trim-context"])
    inference.main(base + ["-o", str(d / "out_shared_untrimmed"), "--no-trim-context"])
    a, _ = audio_io.load(str(d / "out_shared" / "0_utt.wav"))
    b, _ = audio_io.load(str(d / "out_per_window" / "0_utt.wav"))
    c, _ = audio_io.load(str(d / "out_shared_untrimmed" / "0_utt.wav"))
    assert torch.equal(a, b) and torch.equal(a, c)


def test_realtime_converter_matches_oracle(workdir):
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.realtime import RealtimeConverter
    d, sds, _ = workdir
    lib = synthetic.make_library(1000, 1)
    chunk, bs = 320, 8
    rt = RealtimeConverter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), lib, "cuda", chunk=chunk,
                           buffersize=bs, f0_rate=0.5)
    pcm = (synthetic.make_waveform(chunk * (bs + 3), 62)[0].numpy() * 20000).astype(np.int16)
    begin, end = O.realtime_geometry(chunk, bs)
    assert (begin, end) == (rt.begin_of_output, rt.end_of_output)
    phi, outs, refs = 0, [], []
    for s in range(bs + 3):
        o = rt.step(pcm[s * chunk:(s + 1) * chunk])
        if s < bs:
            assert o is None            # the reference emits nothing until the ring holds > buffersize chunks
            continue
        ring = torch.from_numpy(pcm[(s - bs + 1) * chunk:(s + 1) * chunk].astype(np.float32) / 32768)[None]
        wave, phi = O.realtime_step(sds["content_encoder.pt"], sds["f0_estimator.pt"], sds["decoder.pt"], ring, lib, phi,
                                    begin, end, f0_rate=0.5)
        ref = (wave[0].numpy() * 32768).astype(np.int16)
        c = bs * chunk // 2
        refs.append(ref[c - chunk // 2: c + chunk // 2])
        outs.append(o)
    got, want = np.concatenate(outs).astype(np.float64), np.concatenate(refs).astype(np.float64)
    assert got.shape == want.shape == (3 * chunk,)
    assert np.sqrt(np.mean((got - want) ** 2)) / 32768 < 1e-3


def test_realtime_converter_with_resampling_and_gains_matches_oracle(workdir):
    """the streaming loop at 24 kHz in / 24 kHz out with input and output gains (realtime_inference.py:139-187): device
    resampler on both edges of every step (csrc/audio.hip), gain order as in the reference (resample -> gain on the way in,
    gain -> resample on the way out), centre chunk cut at the output rate"""
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.realtime import RealtimeConverter
    d, sds, _ = workdir
    lib = synthetic.make_library(1000, 1)
    chunk, bs, sr = 480, 12, 24000                       # 20 ms chunks; ring of 5760 samples at 24 kHz = 12 frames at 16 kHz
    rt = RealtimeConverter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), lib, "cuda", chunk=chunk, buffersize=bs,
                           input_sr=sr, output_sr=sr, f0_rate=0.5, pitch=1.0, gain=-2.0, input_gain=3.0)
    pcm = (synthetic.make_waveform(chunk * (bs + 3), 65)[0].numpy() * 12000).astype(np.int16)
    internal_chunk = int(chunk * (16000 / sr))
    center16 = int(internal_chunk * bs) // 2
    begin, end = center16 - internal_chunk // 2, center16 + internal_chunk // 2
    assert (begin, end) == (rt.begin_of_output, rt.end_of_output)
    phi, outs, refs = 0, [], []
    for s in range(bs + 3):
        o = rt.step(pcm[s * chunk:(s + 1) * chunk])
        if s < bs:
            assert o is None
            continue
        ring = torch.from_numpy(pcm[(s - bs + 1) * chunk:(s + 1) * chunk].astype(np.float32) / 32768)[None]
        x = O.gain(O.resample(ring, sr, 16000), 3.0)
        wave, phi = O.realtime_step(sds["content_encoder.pt"], sds["f0_estimator.pt"], sds["decoder.pt"], x, lib, phi,
                                    begin, end, f0_rate=0.5, pitch_shift=1.0)
        y = O.resample(O.gain(wave, -2.0), 16000, sr)[0]
        ref = (y.numpy() * 32768).astype(np.int16)
        c = bs * chunk // 2
        refs.append(ref[c - chunk // 2: c + chunk // 2])
        outs.append(o)
    got, want = np.concatenate(outs).astype(np.float64), np.concatenate(refs).astype(np.float64)
    assert got.shape == want.shape == (3 * chunk,)
    assert np.sqrt(np.mean((got - want) ** 2)) / 32768 < 1e-3


def test_realtime_cli_replays_a_hipgraph_with_the_same_samples(workdir):
    """realtime_inference.py (BASELINE config 5: -c 160 -b 16, hipGraph-captured per-chunk pipeline) streaming a 24 kHz file:
    the default (captured graph) and --no-graph runs write identical files"""
    import realtime_inference as rti
    d, _, _ = workdir
    base = ["-dep", str(d / "decoder.pt"), "-cep", str(d / "content_encoder.pt"), "-f0ep", str(d / "f0_estimator.pt"),
            "-lib", str(d / "voice_library.pt"), "-d", "cuda", "-c", "160", "-b", "16", "-f0", "0.5",
            "--input-wav", str(d / "inputs" / "utt.wav")]
    rti.main(base + ["--output-wav", str(d / "rt_graph.wav")])
    rti.main(base + ["--output-wav", str(d / "rt_eager.wav"), "--no-graph"])
    a, sra = audio_io.load(str(d / "rt_graph.wav"))
    b, _ = audio_io.load(str(d / "rt_eager.wav"))
    assert sra == 16000 and a.shape[1] >= 160 * 50 and torch.equal(a, b)


@pytest.mark.parametrize("graph,bs", [(False, 26), (True, 26), (False, 70), (False, 34), (True, 44), (False, 60)])
def test_realtime_interior_reuse_equals_full_recomputation(graph, bs):
    """a ring of 78 frames advancing by 3 frames per step.  With interior reuse only the two edge blocks of
    the ring go through spectrogram / content encoder / f0 estimator / kNN again; the emitted samples must be bitwise those
    of the step that recomputes the whole ring (realtime_inference.py:130-167), eagerly and from the captured hipGraph."""
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.realtime import RealtimeConverter
    lib = synthetic.make_library(3000, 1)
    chunk, steps = 960, 7                    # bs 26: ring of 78 frames (streaming kernels, 33-frame edge blocks); 34 / 44 / 60 / 70: rings of
                                             # 102 / 132 / 180 / 210 frames (plane GEMMs: the 33-frame blocks run as batches of three rows) -- the
                                             # 96 .. 195-frame rings are the ones round 2 had to refuse
    pcm = (synthetic.make_waveform(chunk * (bs + steps), 64)[0].numpy() * 20000).astype(np.int16)
    outs = {}
    for reuse in (False, True):
        rt = RealtimeConverter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), lib, "cuda", chunk=chunk,
                               buffersize=bs, f0_rate=0.5, pitch=1.0, alpha=0.1, reuse_interior=reuse)
        assert rt.reuse == reuse and rt.frames == 3 * bs
        if graph:
            rt.enable_graph()
        waves = []
        for s in range(steps):                                          # the float waveform of the whole ring, not only its int16 centre
            ring = torch.from_numpy(pcm[s * chunk:(s + bs) * chunk].astype(np.float32) / 32768)[None].to("cuda")
            waves.append(rt.step_device(ring, continues=s > 0).clone())   # the caller vouches for the one-chunk advance
        outs[reuse] = torch.stack(waves)
    assert outs[True].shape[0] == steps and torch.isfinite(outs[True]).all()
    assert torch.equal(outs[True], outs[False])
    with pytest.raises(ValueError):                                     # the reference's default ring (24 frames) is too short for it
        RealtimeConverter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), lib, "cuda", chunk=960, buffersize=8,
                          reuse_interior=True)
    with pytest.raises(ValueError):                                     # 63 frames: too short for two 35-frame edge slices
        RealtimeConverter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), lib, "cuda", chunk=960, buffersize=21,
                          reuse_interior=True)


@pytest.mark.parametrize("graph", [False, True])
def test_realtime_interior_reuse_survives_a_broken_chunk_sequence(graph):
    """the carried-over interior frames are only valid when the ring is the previous one advanced by one chunk.
    `step_device` without `continues=True` (a dropped chunk, an unrelated ring) and `reset()` (a second stream on the same
    converter) must fall back to the full front end: results bitwise those of a converter without interior reuse."""
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.realtime import RealtimeConverter
    lib = synthetic.make_library(3000, 1)
    chunk, bs = 960, 26
    pcm = (synthetic.make_waveform(chunk * (bs + 12), 65)[0].numpy() * 20000).astype(np.int16)
    starts = [0, 1, 2, 5, 6, 3]                                          # chunk 3, 4 dropped, then a jump backwards
    outs = {}
    for reuse in (False, True):
        rt = RealtimeConverter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), lib, "cuda", chunk=chunk,
                               buffersize=bs, reuse_interior=reuse)
        if graph:
            rt.enable_graph()
        waves, prev = [], None
        for s in starts:
            ring = torch.from_numpy(pcm[s * chunk:(s + bs) * chunk].astype(np.float32) / 32768)[None].to("cuda")
            waves.append(rt.step_device(ring, continues=(prev is not None and s == prev + 1)).clone())
            prev = s
        # a second stream through step(): reset() empties the ring, the phase and the caches
        rt.reset()
        second = [rt.step(pcm[(4 + i) * chunk:(5 + i) * chunk]) for i in range(bs + 3)]
        assert all(o is None for o in second[:bs]) and all(o is not None for o in second[bs:])
        outs[reuse] = (torch.stack(waves), np.concatenate(second[bs:]))
    assert torch.equal(outs[True][0], outs[False][0])
    assert np.array_equal(outs[True][1], outs[False][1])


def test_realtime_interior_reuse_matches_oracle():
    """row f4 against the CPU oracle (not only against the build's own full recomputation): a 78-frame ring advancing by
    3 frames, 5 emitted chunks, interior reuse on -- int16 RMS / 32768 < 1e-3 like every other streaming test"""
    from module import schema
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.realtime import RealtimeConverter
    lib = synthetic.make_library(1000, 1)
    sds = [synthetic.make_state_dict(s, 2, p) for s, p in ((schema.content_encoder_schema(), "ce."),
                                                          (schema.f0_estimator_schema(), "pe."), (schema.decoder_schema(), "dec."))]
    chunk, bs, emit = 960, 26, 5
    rt = RealtimeConverter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), lib, "cuda", chunk=chunk,
                           buffersize=bs, f0_rate=0.5, reuse_interior=True)
    pcm = (synthetic.make_waveform(chunk * (bs + emit), 66)[0].numpy() * 20000).astype(np.int16)
    begin, end = O.realtime_geometry(chunk, bs)
    phi, outs, refs = 0, [], []
    for s in range(bs + emit):
        o = rt.step(pcm[s * chunk:(s + 1) * chunk])
        if s < bs:
            continue
        ring = torch.from_numpy(pcm[(s - bs + 1) * chunk:(s + 1) * chunk].astype(np.float32) / 32768)[None]
        wave, phi = O.realtime_step(sds[0], sds[1], sds[2], ring, lib, phi, begin, end, f0_rate=0.5)
        ref = (wave[0].numpy() * 32768).astype(np.int16)
        c = bs * chunk // 2
        refs.append(ref[c - chunk // 2: c + chunk // 2])
        outs.append(o)
    got, want = np.concatenate(outs).astype(np.float64), np.concatenate(refs).astype(np.float64)
    assert got.shape == want.shape == (emit * chunk,)
    assert np.sqrt(np.mean((got - want) ** 2)) / 32768 < 1e-3


def test_realtime_rejects_rings_shorter_than_five_frames(workdir):
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.realtime import RealtimeConverter
    with pytest.raises(ValueError):
        RealtimeConverter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), synthetic.make_library(16, 1), "cuda",
                          chunk=160, buffersize=8)


def test_generate_voice_library_writes_reference_format(workdir):
    import generate_voice_library as gvl
    from module.voice_library import VoiceLibrary
    d, sds, _ = workdir
    os.makedirs(d / "corpus", exist_ok=True)
    audio_io.save(str(d / "corpus" / "a.wav"), synthetic.make_waveform(16000 * 6, 7) * 0.4, 16000)
    gvl.main([str(d / "corpus"), "-lib", str(d / "vl_out.pt"), "-cep", str(d / "content_encoder.pt"), "--seed", "3",
              "--num-tokens", "512", "--frames-per-clip", "4"])
    sd = torch.load(d / "vl_out.pt")
    assert list(sd.keys()) == ["tokens"] and tuple(sd["tokens"].shape) == (1, 768, 512) and sd["tokens"].dtype == torch.float32
    # the first slots hold content-encoder frames of the corpus
    wf = synthetic.make_waveform(16000 * 6, 7) * 0.4
    wf = wf / wf.abs().max()
    clips = wf[0, : (wf.shape[1] // 7680) * 7680].view(-1, 7680)
    feats = O.content_encoder(sds["content_encoder.pt"], O.spectrogram(clips))        # [n, 768, 24]
    tok = sd["tokens"][0, :, 0]
    d2 = (feats[:, :, :8] - tok.view(1, 768, 1)).pow(2).sum(dim=1)
    assert d2.min().item() < 1e-4 * tok.pow(2).sum().item()
    VoiceLibrary().load_state_dict(sd)


def test_voice_library_dedup_drops_later_near_duplicates():
    import generate_voice_library as gvl
    base = synthetic.gaussian("dd.base", 5, (768, 40))
    toks = torch.cat([base, base[:, :10] * 1.7 + 1e-4 * synthetic.gaussian("dd.n", 6, (768, 10)), base[:, 20:25]], 1)   # 55 frames
    keep = gvl.dedup_mask(toks.to("cuda"), 0.999)
    assert keep[:40].all() and not keep[40:].any()
    assert gvl.dedup_mask(toks.to("cuda"), 1.5).all()                 # nothing is above an impossible threshold


def test_voice_library_dedup_at_scale_equals_the_sequential_definition():
    """60 000 frames with chains of near-duplicates (a copy of a copy of a copy ...): the device passes must reproduce the
    greedy index-order definition, evaluated here row by row on the host from the same neighbour lists"""
    import generate_voice_library as gvl
    from module.common import PackedLibrary
    g = torch.Generator(device="cuda").manual_seed(9)
    base = torch.randn(768, 20000, device="cuda", generator=g)
    parts = [base]
    for gen in range(2):                                           # generation g + 1 copies generation g (chains of length 3)
        parts.append(parts[-1] + 0.02 * torch.randn(768, 20000, device="cuda", generator=g))
    toks = torch.cat(parts, 1)[:, torch.randperm(60000, device="cuda", generator=g)].contiguous()
    thr, k = 0.9995, 8
    keep = gvl.dedup_mask(toks, thr, k=k, chunk=16384)
    val, idx = PackedLibrary(toks).search(toks.unsqueeze(0), k)
    val, idx = val.cpu().numpy(), idx.cpu().numpy()
    want = np.ones(60000, dtype=bool)
    for i in range(60000):
        near = idx[i][(val[i] > thr) & (idx[i] < i) & (idx[i] >= 0)]
        if near.size and want[near].any():
            want[i] = False
    assert np.array_equal(keep.numpy(), want)
    assert 15000 < int(want.sum()) < 45000                          # chains really occur: neither all kept nor all dropped


def test_realtime_graph_capture_equals_eager(workdir):
    """the whole per-step device pipeline captured into one hipGraph replays to the same samples as eager execution"""
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.realtime import RealtimeConverter
    lib = synthetic.make_library(2000, 1)
    chunk, bs = 160, 16                      # 10 ms chunks, 8-frame ring (BASELINE config 5)
    pcm = (synthetic.make_waveform(chunk * (bs + 6), 63)[0].numpy() * 20000).astype(np.int16)
    outs = {}
    for mode in ("eager", "graph"):
        rt = RealtimeConverter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), lib, "cuda", chunk=chunk,
                               buffersize=bs, f0_rate=0.5)
        if mode == "graph":
            rt.enable_graph()
        got = [rt.step(pcm[s * chunk:(s + 1) * chunk]) for s in range(bs + 6)]
        outs[mode] = np.concatenate([o for o in got if o is not None])
    assert outs["eager"].shape == (6 * chunk,)
    assert np.array_equal(outs["eager"], outs["graph"])


def test_bench_world1_rccl_executes_the_sharded_leg():
    """the RCCL code path on the hardware there is.  `--gpus 1 --force-dist` creates a world-size-1 `nccl`
    (= RCCL) process group: communicator init with device_id, barrier / all_reduce on device tensors (fences, max over
    ranks), all_gather_into_tensor of the content features and all_to_all_single of the exact lists in module/sharded.py,
    with one shard -- so that an 8-GPU run is not the first execution of those calls on device memory."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl",
                        "--library-size", "30000", "--utterances", "3", "--seconds", "4", "--steps", "1", "--warmup", "1",
                        "--window-batch", "8", "--legs", "none"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["ranks"] == {"backend": "nccl", "rccl_ranks": 1, "same_device": False}
    sk = d["sharded_knn"]
    assert sk["backend"] == "nccl" and sk["equals_unsharded"] is True and sk["shards"] == 1, sk
    assert sk["rows_per_shard"] == [30000] and sk["exchange"].startswith("all_gather frames")


def test_bench_two_ranks_gloo_same_device():
    """`python bench.py --gpus 2` launches its own ranks (torch.distributed.run) and prints ONE JSON line; with
    --backend gloo --same-device both ranks share the box's single GPU, which exercises the whole multi-rank path: weak-scaling
    headline (max over ranks) and BASELINE config 4 end to end -- feature all-gather, library-sharded match, list all-gather,
    merge -- checked on the device against the unsharded search and the replicated path (bitwise)."""
    import json
    import subprocess
    env = dict(os.environ, ALIVE_STREAMS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
                        "--library-size", "30000", "--utterances", "3", "--seconds", "4", "--steps", "1", "--warmup", "1",
                        "--window-batch", "8"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["ranks"] == {"backend": "gloo", "gloo_ranks": 2, "same_device": True}
    sk = d["sharded_knn"]
    assert sk["equals_unsharded"] is True and sk["lists_equal_unsharded"] and sk["waveforms_equal_replicated"], sk
    assert sk["shards"] == 2 and sk["rows_per_shard"] == [15000, 15000] and sum(sk["windows_per_rank"]) == sk["global_windows"]
    assert sk["exchange_bytes_received_per_rank"]["frames_allgather_received"] > 0
