"""BASELINE config 3 at batch scale through the whole path: 128 and 384 windows in one `convert_windows`, window batches of
64 / 128 on three side streams (9 GB of scratch per stream), a 200 000-vector library -- against per-window calls
(bitwise) and against the CPU oracle on sampled windows (waveform RMS < 1e-3).  Reference loop: inference.py:96-135."""
import pytest
import torch

import alive_oracle as O
import bench
from module import schema, synthetic

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
M = 200_000


@pytest.fixture(scope="module")
def rig():
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.pipeline import Converter
    g = torch.Generator(device=DEV).manual_seed(200)
    tokens = torch.randn(1, 768, M, device=DEV, generator=g)
    conv = Converter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), DEV).set_library(tokens)
    windows = bench.synth_windows(64, 10.0, 48000, DEV, seed=300)                   # 384 windows x 144 000 samples
    return conv, tokens, windows


@pytest.mark.parametrize("n_win,window_batch,streams", [(128, 64, "3"), (128, 128, "3"), (384, 128, "3"), (384, 64, "3"),
                                                        (384, 128, "1"), (130, 64, "2")])
def test_batched_conversion_equals_per_window_calls(rig, monkeypatch, n_win, window_batch, streams):
    """a window's waveform must not depend on how many windows share its launch, on the batch it falls into or on the side
    stream that batch runs on (130 windows: a ragged last batch of 2)"""
    conv, _, windows = rig
    monkeypatch.setenv("ALIVE_STREAMS", streams)
    w = windows[:n_win]
    out = conv.convert_windows(w, k=4, alpha=0.1, window_batch=window_batch)
    assert torch.isfinite(out).all()
    monkeypatch.setenv("ALIVE_STREAMS", "1")
    for i in sorted({0, 1, window_batch - 1, window_batch % n_win, n_win // 2, n_win - 2, n_win - 1}):
        single = conv.convert_windows(w[i:i + 1].contiguous(), k=4, alpha=0.1, window_batch=1)
        assert torch.equal(single[0], out[i]), f"window {i} differs between the batched and the single launch"


def test_repeated_batched_conversion_is_deterministic(rig, monkeypatch):
    """the same 384-window step five times on three streams: bitwise the same waveforms (scratch reuse across streams,
    weight tables shared by the streams)"""
    conv, _, windows = rig
    monkeypatch.setenv("ALIVE_STREAMS", "3")
    first = conv.convert_windows(windows, k=4, window_batch=128)
    for _ in range(4):
        assert torch.equal(conv.convert_windows(windows, k=4, window_batch=128), first)


def test_batched_conversion_against_the_oracle(rig, monkeypatch):
    """three windows of the 384-window batch through the CPU oracle with the same 200 k-vector library:
    waveform RMS error < 1e-3 (the north-star bar), f0 identical"""
    conv, tokens, windows = rig
    monkeypatch.setenv("ALIVE_STREAMS", "3")
    out = conv.convert_windows(windows, k=4, window_batch=128).cpu()
    cpu = [synthetic.make_state_dict(s, 2, p) for s, p in ((schema.content_encoder_schema(), "ce."),
                                                          (schema.f0_estimator_schema(), "pe."),
                                                          (schema.decoder_schema(), "dec."))]
    lib = tokens.cpu()
    torch.set_num_threads(16)
    for i in (0, 200, 383):
        ref = O.convert_window(cpu[0], cpu[1], cpu[2], windows[i:i + 1].cpu(), lib, k=4, alpha=0.0)
        err = (out[i:i + 1] - ref).pow(2).mean().sqrt().item()
        assert err < 1e-3, f"window {i}: RMS error {err:.3e} against the oracle (signal RMS {ref.pow(2).mean().sqrt().item():.3f})"


def test_match_at_200k_vectors_against_brute_force(rig):
    """the kNN half of config 3: all 172 800 frames of the batch against the 200 k-vector library, 12 000 of them checked
    against a brute-force fp32 scan"""
    conv, tokens, windows = rig
    feat = torch.cat([conv.features(windows[i:i + 128])[0] for i in range(0, 384, 128)], 0)
    val, idx = conv.library.search(feat, 4)
    assert tuple(val.shape) == (172_800, 4)
    flat = feat.permute(0, 2, 1).reshape(-1, 768)
    sel = torch.randperm(flat.shape[0], device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))[:12_000]
    q = flat[sel]
    qn = q / q.norm(dim=1, keepdim=True)
    ln = tokens[0] / tokens[0].norm(dim=0, keepdim=True)
    top = torch.topk(qn @ ln, 5, dim=1)
    safe = (top.values[:, 3] - top.values[:, 4]) > 1e-5
    got = torch.sort(idx[sel].long(), dim=1).values[safe]
    want = torch.sort(top.indices[:, :4], dim=1).values[safe]
    assert int(safe.sum()) > 11_000 and torch.equal(got, want)
    assert float((val[sel] - top.values[:, :4]).abs().max()) <= 2e-6


@pytest.mark.parametrize("n_utt,window_batch", [(8, 64), (64, 128)])
def test_overlap_sharing_equals_the_per_window_front_end(rig, monkeypatch, n_utt, window_batch):
    """the windows of an utterance overlap by two thirds (inference.py:94-101): with share_overlap the spectrogram, the f0
    estimator, the content encoder and the kNN match run once per utterance (+ the two edge blocks of every window) and the
    windows are assembled by copies -- every waveform sample bitwise as before, ~51 % of the frames through the match"""
    conv, _, windows = rig
    monkeypatch.setenv("ALIVE_STREAMS", "3")
    w = windows[:n_utt * 6]
    ref = conv.convert_windows(w, k=4, alpha=0.1, pitch_shift=1.0, intonation=1.2, window_batch=window_batch)
    got = conv.convert_windows(w, k=4, alpha=0.1, pitch_shift=1.0, intonation=1.2, window_batch=window_batch, share_overlap=6)
    assert torch.equal(got, ref)
    assert conv.last_front_end_frames == n_utt * (8 * 150 + 6 * 28)
    assert conv.last_front_end_frames < 0.52 * w.shape[0] * 450


def test_overlap_sharing_of_one_utterance(rig):
    """Converter.convert(share_overlap=True): odd window counts, short chunks (50 frames per chunk), a chunk that is too short
    falls back to the per-window front end"""
    conv, _, _ = rig
    wf = (0.3 * synthetic.make_waveform(16000 * 7 + 123, 41)).to(DEV)
    for chunk in (48000, 16000, 4800):
        assert torch.equal(conv.convert(wf, chunk=chunk, k=4, share_overlap=True), conv.convert(wf, chunk=chunk, k=4))
    # "auto" (the CLI's setting): 7 s at chunk 16000 = 10 windows against 200 k vectors stays per window, 60 s is shared
    conv.last_front_end_frames = None
    conv.convert(wf, chunk=16000, k=4, share_overlap="auto")
    assert conv.last_front_end_frames is None
    long = (0.3 * synthetic.make_waveform(16000 * 60, 42)).to(DEV)
    out = conv.convert(long, chunk=16000, k=4, share_overlap="auto")
    assert conv.last_front_end_frames is not None and torch.equal(out, conv.convert(long, chunk=16000, k=4))


@pytest.mark.parametrize("n_utt,window_batch", [(8, 64), (64, 128)])
def test_overlap_sharing_with_context_trimming_keeps_the_same_samples(rig, monkeypatch, n_utt, window_batch):
    """sharing and trimming together (the CLI's default since round 5): front end once per utterance, kNN match on the
    union of the windows' trimmed ranges only, decoder on frames [cf - 32, 2 cf + 16) of every window -- the kept centre
    third is bitwise that of the plain path (inference.py:96-101,132-135)"""
    conv, _, windows = rig
    monkeypatch.setenv("ALIVE_STREAMS", "3")
    w = windows[:n_utt * 6]
    kw = dict(k=4, alpha=0.1, pitch_shift=1.0, intonation=1.2, window_batch=window_batch)
    ref = conv.convert_windows(w, **kw)
    got = conv.convert_windows(w, share_overlap=6, keep_frames=(150, 300), **kw)
    assert torch.equal(got[:, 48000:96000], ref[:, 48000:96000])
    # per signal: frames [118, 5 * 150 + 316) go through the match -- 35 % of the per-window path's frames
    assert conv.last_front_end_frames == n_utt * (5 * 150 + 316 - 118)
    trim_only = conv.convert_windows(w, keep_frames=(150, 300), **kw)
    assert torch.equal(got[:, 48000:96000], trim_only[:, 48000:96000])


def test_shared_and_trimmed_conversion_of_one_utterance(rig):
    """Converter.convert(share_overlap=..., trim_context=True) == the plain path, bitwise, over chunk sizes (a chunk of 15
    frames cannot be trimmed onto interior frames: the code falls back to trimming / sharing alone) and an odd length"""
    conv, _, _ = rig
    wf = (0.3 * synthetic.make_waveform(16000 * 7 + 123, 41)).to(DEV)
    for chunk in (48000, 16000, 4800):
        ref = conv.convert(wf, chunk=chunk, k=4)
        assert torch.equal(conv.convert(wf, chunk=chunk, k=4, share_overlap=True, trim_context=True), ref)
        assert torch.equal(conv.convert(wf, chunk=chunk, k=4, trim_context=True), ref)
    long = (0.3 * synthetic.make_waveform(16000 * 60, 42)).to(DEV)
    conv.last_front_end_frames = None
    out = conv.convert(long, chunk=16000, k=4, share_overlap="auto", trim_context=True)
    assert conv.last_front_end_frames is not None                   # shared (62 windows against 200 k vectors) ...
    assert conv.last_front_end_frames < 0.4 * 62 * 150              # ... and trimmed: about a third of the per-window frames
    assert torch.equal(out, conv.convert(long, chunk=16000, k=4))


def test_a_short_tail_of_edge_blocks_stays_on_the_batch_kernels(rig):
    """the shared front end runs the windows' edge blocks 256 windows at a time; 257 - 259 windows used to leave a tail of
    1 - 3 windows = 30 - 90 frame columns, below the 96 from which the plane GEMMs run -- those windows' f0 / features then came from
    the fp32 streaming kernels, which round differently, and the stitched file was no longer bitwise the per-window path's.  The tail
    is folded into the chunk before it (module/pipeline.py::_edge_chunks): one utterance of 257 windows, shared + trimmed and shared
    alone, against the plain path."""
    from module.pipeline import _edge_chunks
    assert _edge_chunks(257, 256) == [(0, 257)] and _edge_chunks(260, 256) == [(0, 256), (256, 260)]
    conv, _, _ = rig
    chunk = 46 * 320                                                # 138-frame windows: the shortest whose trimmed range is interior frames
    wf = (0.3 * synthetic.make_waveform(chunk * 255 - 77, 43)).to(DEV)           # 257 windows (make_windows pads by two chunks)
    from module.pipeline import make_windows
    assert make_windows(wf, chunk)[0].shape[0] == 257
    ref = conv.convert(wf, chunk=chunk, k=4)
    assert torch.equal(conv.convert(wf, chunk=chunk, k=4, share_overlap=True), ref)
    assert torch.equal(conv.convert(wf, chunk=chunk, k=4, share_overlap=True, trim_context=True), ref)
