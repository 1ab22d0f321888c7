"""CPU tests: C-ABI exports, host-side logic (windowing, packing, WAV I/O, resampler, CLI flags) and the
multi-GPU protocol on a world_size-2 gloo group.  No GPU compute is called."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import alive_oracle as O
from module import _native as nat
from module import _pack, audio_io, schema, synthetic
from module.pipeline import make_windows, stitch
from module.sharded import ShardedLibrary, partition_windows, shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    hdr = open(os.path.join(ROOT, "include", "alive_vc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(alive_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = ctypes.CDLL(nat.LIB_PATH)
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(nat.PROTOTYPES), declared ^ set(nat.PROTOTYPES)
    assert nat.lib().alive_version() == 1
    assert nat.lib().alive_library_padded_rows(1000) == 1024


def test_argument_errors_are_reported_without_a_gpu():
    L = nat.lib()
    assert L.alive_library_pack(None, 10, 768, None, None, None, None) == -1
    assert b"null" in L.alive_last_error()
    rc = L.alive_knn_search(1, 1, 5, 1, 1, 1, 3, 0, 4, 1, 1, 1, None)     # M < k
    assert rc == -1 and b"fewer than k" in L.alive_last_error()
    rc = L.alive_decoder_forward(1, 1, 1, None, 0, 0, 1, 4, 1, None, 1, None)    # Lf < 5
    assert rc == -1 and b"at least 5 frames" in L.alive_last_error()


def test_knn_workspace_query_covers_every_search_entry_point():
    """alive_knn_search_strict takes no workspace size, so the general query must be the bound that covers its layout (a C
    caller built against an older header sizes ONE buffer with alive_knn_workspace_bytes); the tight size has a name of its own"""
    L = nat.lib()
    for tt, m in ((8, 1000), (450, 50000), (172800, 1000000)):
        gen, strict, fast = (int(f(tt, m)) for f in (L.alive_knn_workspace_bytes, L.alive_knn_workspace_bytes_strict,
                                                     L.alive_knn_workspace_bytes_fast))
        assert gen == strict and strict >= fast + 2 * 2 * 768 * tt > fast > 0


def test_product_path_refuses_cpu_tensors():
    from module.common import match_features
    from module.content_encoder import ContentEncoder
    with pytest.raises(RuntimeError):
        match_features(torch.zeros(1, 768, 4), torch.zeros(1, 768, 8))
    with pytest.raises(RuntimeError):
        ContentEncoder(seed=2)(torch.zeros(1, 641, 4))
    assert "oracle" not in " ".join(sys.modules[m].__file__ or "" for m in list(sys.modules)
                                    if m.startswith("module.") and hasattr(sys.modules[m], "__file__"))


def test_weight_tables_cover_the_schema():
    for mid, (sch, pk) in enumerate([(schema.content_encoder_schema(), _pack.pack_content_encoder),
                                     (schema.f0_estimator_schema(), _pack.pack_f0_estimator),
                                     (schema.decoder_schema(), _pack.pack_decoder)]):
        packed = pk(synthetic.make_state_dict(sch, 2))
        assert set(packed) == set(nat.weight_names(mid))
        for k, v in packed.items():
            assert v.dtype in (torch.float32, torch.bfloat16) and v.is_contiguous()
            if k.endswith(".W") and v.dtype == torch.float32 and v.dim() == 2:   # exact-fp32 MFMA kernel: [Co_pad16][K_pad16]
                assert v.shape[0] % 16 == 0 and v.shape[1] % 16 == 0
            if k.endswith(".W") and v.dtype == torch.bfloat16:         # split kernel, k-blocked: [planes][KW * Ci_pad32 / 32][Co_pad16][32]
                # encoders: 3 bf16 planes; decoder: 2, or 2 + the fp16 slab of the plain kernels (decoder precision mode 1, _pack.pack_conv_split_h)
                plain = mid == 2 and (k.startswith("flt.blk0.") or k.startswith("flt.blk1.") or k.startswith("flt.up") or k == "flt.mid.W" or k == "fe.normfilm.W"
                                      or (k.startswith("fe.mid") and ".pw" in k))
                f16s = mid < 2 and (".pw" in k or (mid == 1 and k == "output.W"))      # encoders' pointwise convs, the f0 classifier: 3 bf16 planes + the fp16 (hi, lo) pair (_pack.pack_conv_split_f16s)
                assert v.dim() == 4 and v.shape[0] == (5 if f16s else 3 if mid < 2 or plain else 2) and v.shape[2] % 16 == 0 and v.shape[3] == 32, k
            if k.endswith(".ws"):                          # 1 / scale of that pair: a power of two
                assert v.shape == (1,) and v.dtype == torch.float32 and float(torch.log2(v)[0]) == round(float(torch.log2(v)[0]))
    sd = synthetic.make_state_dict(schema.decoder_schema(), 2)
    p = _pack.pack_decoder(sd)
    assert p["flt.in.W"].shape == (56,) and p["flt.down0.W"].shape == (256,) and p["flt.out.W"].shape == (56,)   # streaming edge kernels
    assert p["flt.film.W"].shape == (2, 16, 4128, 32) and p["flt.film.post"].sum().item() == 2064
    w = sd["filter.blocks.0.blocks.1.c2.conv.conv.weight"]                 # [256, 256, 5]
    rows = _pack.unpack_conv_split(p["flt.blk0.1.c2.W"]).float()          # k-blocked -> [planes][co][k]
    hi, lo = rows[0], rows[1]
    back = (hi + lo).view(256, 5, 256).permute(0, 2, 1)                    # tap-major -> [co, ci, j]
    assert (back - w).abs().max().item() <= 2.0 ** -16 * w.abs().max().item()
    h = p["flt.blk0.1.c2.W"][2].view(torch.float16)                         # the third slab: the same weights as ONE fp16 plane, same layout
    hb = h.permute(1, 0, 2).reshape(256, 5, 256).permute(0, 2, 1)
    assert torch.equal(hb, w.half()) and p["flt.down2.Wp"].shape[0] == 3 and p["flt.down3.Wp"].shape[0] == 3
    sdc = synthetic.make_state_dict(schema.content_encoder_schema(), 2)
    pc = _pack.pack_content_encoder(sdc)
    w = sdc["mid_layers.1.pw_conv1.weight"]                                   # [1536, 512, 1]
    W5, ws = pc["mid1.pw1.W"], pc["mid1.pw1.ws"]
    sc = 1.0 / float(ws)
    assert 8192.0 < float(w.abs().max()) * sc <= 16384.0
    hi = W5[3].view(torch.float16).permute(1, 0, 2).reshape(1536, 512).double()
    lo = W5[4].view(torch.float16).permute(1, 0, 2).reshape(1536, 512).double()
    assert ((hi + lo) / sc - w[:, :, 0].double()).abs().max().item() <= 2.0 ** -21 * float(w.abs().max())     # 22 bits of the largest element
    w = torch.arange(2 * 3 * 4, dtype=torch.float32).view(2, 3, 4)            # ConvT [Ci=2, Co=3, r=4]
    W, b = _pack.pack_convT(w, torch.tensor([1.0, 2.0, 3.0]))
    assert W[1 * 4 + 2, 1].item() == w[1, 1, 2].item() and b.tolist() == [1.0] * 4 + [2.0] * 4 + [3.0] * 4


@pytest.mark.parametrize("L", [1, 15999, 16000, 48000, 100001])
def test_windowing_equals_oracle(L):
    wf = synthetic.make_waveform(L, 60)
    w, total = make_windows(wf, 48000)
    ow, ot = O.make_windows(wf, 48000)
    assert total == ot and torch.equal(w, ow)
    assert stitch(w, total, 48000).shape == (1, L)
    assert torch.equal(stitch(w, total, 48000), wf)          # centre thirds tile the input exactly


def test_wav_roundtrip_and_resampler(tmp_path):
    x = synthetic.make_waveform(24000, 5) * 0.3
    for enc, tol in (("float32", 0.0), ("pcm16", 1.0 / 32768)):
        p = str(tmp_path / f"a_{enc}.wav")
        audio_io.save(p, x, 24000, enc)
        y, sr = audio_io.load(p)
        assert sr == 24000 and y.shape == x.shape and (y - x).abs().max().item() <= tol
    assert audio_io.resample(x, 16000, 16000) is x
    t = torch.arange(24000, dtype=torch.float64) / 24000
    tone = torch.sin(2 * np.pi * 440 * t).float()[None]
    down = O.resample(tone, 24000, 16000)                 # oracle restatement (the product resampler is csrc/audio.hip)
    assert down.shape == (1, 16000)
    ref = torch.sin(2 * np.pi * 440 * torch.arange(16000, dtype=torch.float64) / 16000).float()[None]
    assert (down - ref)[:, 200:-200].abs().max().item() < 2e-3
    up = O.resample(down, 16000, 24000)
    assert up.shape == (1, 24000) and (up - tone)[:, 300:-300].abs().max().item() < 4e-3
    assert abs(audio_io.gain(torch.ones(1), 1.0).item() - 10 ** 0.05) < 1e-6


def test_cli_flags_match_the_reference():
    sys.path.insert(0, os.path.join(ROOT, "alive-vc_amd"))
    import importlib
    inf = importlib.import_module("inference").build_parser()
    d = vars(inf.parse_args([]))
    assert (d["inputs"], d["outputs"], d["decoder_path"], d["chunk"], d["k"], d["gain"], d["alpha"], d["f0_rate"],
            d["intonation"], d["pitch"], d["voice_library_path"], d["target"]) == \
        ("./inputs/", "./outputs/", "decoder.pt", 48000, 4, 1.0, 0.0, 1.0, 1.0, 0, "NONE", "NONE")
    # -d / --device: the reference defaults to "cpu" (inference.py:31, realtime_inference.py:23-24).  This package has no CPU path by
    # contract (a missing GPU raises instead of falling back), so its documented deviation is the default "cuda" -- the flag, its
    # long name and its choices are the reference's
    assert d["device"] == "cuda" and "-d" in inf._option_string_actions and "--device" in inf._option_string_actions
    # this build's own switches: by default the CLI trims each window's work to what reaches the kept centre third and shares the
    # front end between overlapping windows (bitwise the same file); each has an off switch, --trim-context stays accepted
    assert not d["no_trim_context"] and not d["no_share_overlap"]
    for flag in ("--trim-context", "--no-trim-context", "--no-share-overlap", "--knn-strict", "--window-batch", "--pcm16"):
        assert flag in inf._option_string_actions, flag
    rt = importlib.import_module("realtime_inference").build_parser()
    d = vars(rt.parse_args([]))
    assert d["device"] == "cuda" and set(rt._option_string_actions["-d"].choices) == {"cpu", "cuda", "mps"}
    assert (d["buffersize"], d["chunk"], d["input_sr"], d["output_sr"], d["gain"], d["k"]) == (8, 960, 16000, 16000, 0.0, 4)
    for flag in ("-dep", "-cep", "-f0ep", "-f0", "-p", "-t", "-a", "-lib", "-wpe", "-isr", "-osr", "-lsr", "-ic", "-oc", "-lc",
                 "-ig", "-b", "-c", "-l", "-fp16"):
        assert flag in rt._option_string_actions, flag


def test_shard_bounds_and_window_partition():
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(1_000_000, 8)[7] == (875000, 1000000)
    cover = []
    for r in range(3):
        s = partition_windows(384, 3, r)
        cover += list(range(384))[s]
    assert cover == list(range(384))


def _cpu_search_factory(lib_DxM, begin, end):
    """oracle-arithmetic stand-in for the HIP search of one slab (test only)."""
    slab = lib_DxM[:, begin:end]

    def search(source, k):
        s = source.transpose(1, 2)
        r = slab.t().unsqueeze(0).expand(s.shape[0], -1, -1)
        cos = torch.bmm(s / torch.norm(s, dim=2, keepdim=True), (r / torch.norm(r, dim=2, keepdim=True)).transpose(1, 2))
        top = torch.topk(cos, k, dim=2)
        return top.values.reshape(-1, k).contiguous(), (top.indices.reshape(-1, k) + begin).to(torch.int32).contiguous()
    return search


def _cpu_merge_factory(lib_DxM):
    def merge(gv, gi, S, k, alpha, source, return_indices=False):
        n, d, t = source.shape
        v = gv.permute(1, 0, 2).reshape(n * t, S * k)
        i = gi.permute(1, 0, 2).reshape(n * t, S * k).long()
        best = torch.topk(v, k, dim=1).indices
        sel = torch.gather(i, 1, best)                                    # [Tt, k]
        picked = lib_DxM.t()[sel].mean(dim=1).view(n, t, d).transpose(1, 2)
        out = picked * (1 - alpha) + source * alpha
        return (out, sel) if return_indices else out
    return merge


def _worker(rank, world, port, q, n5=5, t5=9):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = synthetic.make_library(777, 31)[0]
        src = synthetic.gaussian("sh.src", 32, (2, 768, 21))
        b, e = shard_bounds(777, world)[rank]
        sl = ShardedLibrary(_cpu_search_factory(lib, b, e), _cpu_merge_factory(lib))
        out = sl.match(src, k=4, alpha=0.25)
        ref = O.match_features(src, lib.unsqueeze(0).expand(2, -1, -1), 4, 0.25)
        # data-parallel windows: every rank converts its own slice; gather on rank 0 reproduces the whole
        mine = partition_windows(2, world, rank)
        parts = [None] * world
        dist.all_gather_object(parts, out[mine])
        whole = torch.cat(parts, 0)
        # BASELINE config 4 end to end: frames are data-parallel too (uneven window counts: 3 + 2), so the protocol
        # all-gathers the features first, every rank searches ALL frames in its slab, and merges only its own
        src5 = synthetic.gaussian("sh.src5", 33, (n5, 768, t5))
        counts = [e_ - b_ for b_, e_ in shard_bounds(n5, world)]
        own = src5[partition_windows(n5, world, rank)]
        got, gidx = sl.match_distributed(own, k=3, alpha=0.1, counts=counts, return_indices=True)
        ref5, ridx5, _ = O.match_features(src5, lib.unsqueeze(0).expand(n5, -1, -1), 3, 0.1, return_indices=True)
        want = ref5[partition_windows(n5, world, rank)]
        widx = ridx5[partition_windows(n5, world, rank)].reshape(-1, 3)
        e3 = float((got - want).abs().max())
        same_idx = bool((torch.sort(gidx, 1).values == torch.sort(widx, 1).values).all())
        xb = sl.last_exchange_bytes
        n_max = max(counts)
        ok_bytes = xb["frames_allgather_received"] == (world - 1) * n_max * 768 * t5 * 4 and \
            xb["lists_alltoall_received"] == (world - 1) * n_max * t5 * 3 * 8   # routed: only this rank's frames arrive
        q.put((rank, float((out - ref).abs().max()), float((whole - ref).abs().max()), e3, same_idx and ok_bytes))
    finally:
        dist.destroy_process_group()


def _run_world(world, port_base, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = port_base + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, e1, e2, e3, ok in res:
        assert e1 < 1e-5 and e2 < 1e-5 and e3 < 1e-5 and ok, res


def test_sharded_knn_protocol_world2_gloo():
    _run_world(2, 29500)


def test_sharded_knn_protocol_world8_gloo_uneven():
    """BASELINE config 4's shape: 8 ranks, a 777-row library (slabs of 98 and 97 rows), 50 windows (counts 7, 7, 6, ...: the
    short ranks pad their all-gather contribution, the routed lists carry only the owner's frames)"""
    assert [e - b for b, e in shard_bounds(777, 8)] == [98] + [97] * 7
    assert [e - b for b, e in shard_bounds(50, 8)] == [7, 7, 6, 6, 6, 6, 6, 6]
    _run_world(8, 33500, 50, 3)


def test_overlap_sharing_geometry_against_the_oracle():
    """The index arithmetic of Converter.features_shared (module/pipeline.py) restated with the CPU oracle's networks: content
    frames of a window that are >= EDGE frames from its edges equal the frames of ONE pass over the padded signal, and the
    edge frames equal those of a 30-frame slice of the window -- i.e. EDGE = 12 (ConvNeXt context) + 2 (STFT reflect padding)
    and NET_MARGIN = 16 are enough for the REFERENCE's arithmetic, not only for this build's kernels."""
    from module.pipeline import EDGE, NET_MARGIN, SPEC_MARGIN
    ce = synthetic.make_state_dict(schema.content_encoder_schema(), 2, "ce.")
    chunk = 16000                                                     # 50 frames per chunk, windows of 150 frames
    wf = 0.3 * synthetic.make_waveform(16000 * 3 + 77, 71)
    windows, _ = make_windows(wf, chunk)                              # [n, 48000]
    n, L = windows.shape
    lf, cf = L // 320, chunk // 320
    per_window = torch.cat([O.content_encoder(ce, O.spectrogram(windows[i:i + 1])) for i in range(n)], 0)      # the reference's order
    sig = torch.cat([windows[:, :chunk].reshape(1, n * chunk), windows[-1:, chunk:]], dim=1)                  # padded signal, (n + 2) chunks
    assert sig.shape[1] == (n + 2) * chunk
    whole = O.content_encoder(ce, O.spectrogram(sig))
    nl = EDGE + NET_MARGIN
    for g in range(n):
        shared = torch.empty(1, 768, lf)
        shared[:, :, EDGE:lf - EDGE] = whole[:, :, g * cf + EDGE:g * cf + lf - EDGE]
        w = windows[g:g + 1]
        left = O.content_encoder(ce, O.spectrogram(w[:, :(nl + SPEC_MARGIN) * 320])[:, :, :nl])
        right = O.content_encoder(ce, O.spectrogram(w[:, L - (nl + SPEC_MARGIN) * 320:])[:, :, SPEC_MARGIN:])
        shared[:, :, :EDGE] = left[:, :, :EDGE]
        shared[:, :, lf - EDGE:] = right[:, :, NET_MARGIN:]
        torch.testing.assert_close(shared, per_window[g:g + 1], rtol=1e-4, atol=1e-5)
    # ... while frames near the window edge do differ from the whole-signal pass (they see the window's own padding): the edge
    # blocks are needed
    g = 1
    near = whole[:, :, g * cf + 4] - per_window[g, :, 4]
    assert near.abs().max() > 1e-3


def test_filter_mid_is_built_with_the_flags_its_schedule_needs():
    """csrc/Makefile must compile filter_mid.hip with -fno-slp-vectorize (packed fp32 math beside MFMAs costs more than it saves, and
    SLP is what made the 16x16x32 kernel of rounds 1-3 non-deterministic) and with the max-ilp scheduling strategy (the default one
    serialises the four dependency chains of an epilogue stage); the source refuses to compile without the macro that accompanies
    the flags (DESIGN.md 3.2b')"""
    mk = open(os.path.join(ROOT, "alive-vc_amd", "csrc", "Makefile")).read()
    line = [ln for ln in mk.splitlines() if ln.startswith("FLAGS_filter_mid")]
    assert len(line) == 1 and "-fno-slp-vectorize" in line[0] and "-DALIVE_FILTER_MID_NO_SLP" in line[0], line
    assert "-amdgpu-sched-strategy=max-ilp" in line[0], line
    small = [ln for ln in mk.splitlines() if ln.startswith("FLAGS_filter_small")]           # same design since round 4, same flags
    assert len(small) == 1 and "$(FLAGS_filter_mid)" in small[0], small
    assert "#ifndef ALIVE_FILTER_MID_NO_SLP" in open(os.path.join(ROOT, "alive-vc_amd", "csrc", "filter_small.hip")).read()
    assert "$(FLAGS_$*)" in mk
    src = open(os.path.join(ROOT, "alive-vc_amd", "csrc", "filter_mid.hip")).read()
    assert "#ifndef ALIVE_FILTER_MID_NO_SLP" in src and "#error" in src
    assert "v_mfma_f32_16x16x32" not in src and "mfma_f32_16x16x32_bf16(" not in src          # the shape that carried the defect is gone
    assert "ALIVE_CHAIN_GAP(15);" in src and "ALIVE_FB64_NO_CHAIN_GAP" in src                   # the wait states between two chains
    assert "diag.hip" not in [w for ln in mk.splitlines() if ln.startswith("SRCS") for w in ln.split()]      # measurement kernels stay out of the product .so


def test_no_listing_starts_an_mfma_chain_beside_an_unread_accumulator_without_the_gap(tmp_path):
    """DESIGN.md 3.2b' (the accumulation-chain hazard): every shipped MFMA kernel's gfx950 listing is scanned for the shape that
    produced wrong last accumulator registers in the fused 64-channel FilterBlock -- the first MFMA of a new chain issued while
    a finished chain's accumulator is still unread -- and each such chain switch must carry >= 8 idle wait states (common.h
    ALIVE_CHAIN_GAP) between the two chains.  The scan must flag the reproducer build (-DALIVE_FB64_NO_CHAIN_GAP) and the
    un-pinned filter_small build, so that the test cannot pass by the scanner going blind."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mfma_hazard_scan as hz
    csrc = os.path.join(ROOT, "alive-vc_amd", "csrc")
    base = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
            "--cuda-device-only", "-S"]
    mk = open(os.path.join(csrc, "Makefile")).read()
    mid = [ln for ln in mk.splitlines() if ln.startswith("FLAGS_filter_mid")][0].split(":=")[1].split()
    jobs = {"knn": ("knn.hip", []), "gemm_planes": ("gemm_planes.hip", []), "conv_split": ("conv_split.hip", []), "conv": ("conv.hip", []),
            "filter_mid": ("filter_mid.hip", mid), "filter_small": ("filter_small.hip", mid),
            "filter_mid_nogap": ("filter_mid.hip", mid + ["-DALIVE_FB64_NO_CHAIN_GAP"]),
            "filter_small_nogap": ("filter_small.hip", mid + ["-DALIVE_FBS_NO_CHAIN_GAP"])}

    def build(item):
        tag, (src, extra) = item
        out = str(tmp_path / f"{tag}.s")
        subprocess.run([hipcc] + base + extra + [os.path.join(csrc, src), "-o", out], check=True, capture_output=True, timeout=900)
        return tag, out
    with ThreadPoolExecutor(4) as ex:
        lst = dict(ex.map(build, jobs.items()))
    seen = {}
    for tag, path in lst.items():
        r = hz.chain_gap_scan(path)
        seen[tag] = (r["mfma"], len(r["switches"]), sum(1 for s_ in r["switches"] if s_[4] < hz.CHAIN_GAP_MIN))
    for tag in ("knn", "gemm_planes", "conv_split", "conv", "filter_mid", "filter_small"):
        assert seen[tag][0] > 100 and seen[tag][2] == 0, (tag, seen)
    # the pattern exists where DESIGN says it does (fused FilterBlocks, fp8 scoring kernel) and only there
    assert seen["filter_mid"][1] >= 20 and seen["filter_small"][1] >= 40 and seen["knn"][1] >= 4, seen
    assert seen["gemm_planes"][1] == seen["conv_split"][1] == seen["conv"][1] == 0, seen
    assert seen["filter_mid_nogap"][2] >= 20 and seen["filter_small_nogap"][2] >= 40, seen
    # round 5: the fp16 forms really run on the fp16 MFMA with the saturating conversion, without spills, at the occupancy DESIGN 3.2d / 3.2e
    # state (two blocks per CU for the GEMMs, three for the plain conv)
    import re

    def kernels(path, pattern):
        txt = open(path).read()
        out = {}
        for m in re.finditer(r"^(_Z\w*" + pattern + r"\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", txt, re.S | re.M):
            body = m.group(2)
            out[m.group(1)] = {"f16": body.count("v_mfma_f32_32x32x16_f16"), "bf16": body.count("v_mfma_f32_32x32x16_bf16"),
                               "cvt": body.count("v_cvt_pk_f16_f32"), "scratch": body.count("scratch_"),
                               "vgpr": int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", body).group(1))}
        return out
    conv = kernels(lst["conv_split"], "conv_split_kernelILi128ELi1ELb1ELb1E")          # <128, 1, PLANES, W22>: the decoder's 256-channel convs
    assert len(conv) == 1, list(conv)
    for k, v in conv.items():
        assert v["f16"] >= 8 and v["bf16"] == 0 and v["cvt"] >= 4 and v["scratch"] == 0 and v["vgpr"] <= 168, (k, v)       # <= 168: three waves per SIMD
    gem = kernels(lst["gemm_planes"], "gemm_planes_kernelILi2ELi2ELi2ELi[0-3]ELb[01]ELb[01]E")
    f16s = {k: v for k, v in gem.items() if k.endswith("Lb0ELb1EEEv9AliveGemmiilliiiNS_8GemmWalkEPx")}         # <2, 2, 2, ACT, false, F16S>
    kb2 = {k: v for k, v in gem.items() if k.endswith("Lb1ELb0EEEv9AliveGemmiilliiiNS_8GemmWalkEPx")}          # <2, 2, 2, ACT, KB2, false>
    assert len(f16s) == 4 and len(kb2) == 3, (list(gem))
    for k, v in list(f16s.items()) + list(kb2.items()):
        assert v["f16"] >= 8 and v["bf16"] == 0 and v["scratch"] == 0 and v["vgpr"] <= 256, (k, v)                # two blocks of four waves per CU
    assert all(v["cvt"] >= 4 for k, v in f16s.items() if "ELi3ELb0ELb1E" not in k)           # (the argmax form writes no planes)


def test_fp8_scoring_kernel_listing_keeps_two_accumulator_sets(tmp_path):
    """the fp8 scoring kernel is 14 % faster because a tile's accumulators stay in a second AGPR set and are copied
    out under the NEXT tile's MFMAs.  hipcc arrived at the slow form by itself once (one set, 64 v_accvgpr_read + 64
    v_accvgpr_write per tile behind a drained matrix pipe), results unchanged, so a compiler update could bring it back
    silently: the listing of the committed source must show four accumulator tuples as MFMA destinations and no
    v_accvgpr_write between the MFMAs of the two tile bodies of the main loop."""
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / "knn.s"
    src = os.path.join(ROOT, "alive-vc_amd", "csrc", "knn.hip")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
                    "--cuda-device-only", "-S", src, "-o", str(out)], check=True, capture_output=True, timeout=600)
    ls = out.read_text().split("\n")
    start = [i for i, l in enumerate(ls) if "knn_score8_kernel" in l and l.startswith("_ZN") and ":" in l][0]
    end = [i for i, l in enumerate(ls) if i > start and ".amdhsa_kernel" in l][0]
    body = ls[start:end]
    mf = [i for i, l in enumerate(body) if "v_mfma_scale_f32_32x32x64_f8f6f4" in l]
    assert len(mf) == 72, len(mf)                        # two tile bodies of the loop + the tail tile, 24 MFMAs each
    dst = {re.search(r"f8f6f4 (a\[\d+:\d+\])", body[i]).group(1) for i in mf[:48]}
    assert dst == {"a[0:15]", "a[16:31]", "a[32:47]", "a[48:63]"}, dst
    moved = [body[i].strip() for i in range(mf[0], mf[47]) if "v_accvgpr_write" in body[i]]
    assert not moved, moved[:4]
    reads = sum("v_accvgpr_read" in body[i] for i in range(mf[0], mf[47]))
    assert reads == 64, reads                            # 32 per tile body: the explicit late copies and nothing else
    # the fp6 kernel of round 5 (three column tiles per wave): three tile-pairs of accumulator tuples, 36 MFMAs per tile body on e2m3
    # operands of SIX registers each (an eight-register operand class would not fit 96 stationary frames), no spills
    start = [i for i, l in enumerate(ls) if "knn_score6_kernel" in l and l.startswith("_ZN") and ":" in l][0]
    end = [i for i, l in enumerate(ls) if i > start and ".amdhsa_kernel" in l][0]
    body = ls[start:end]
    mf = [i for i, l in enumerate(body) if "v_mfma_scale_f32_32x32x64_f8f6f4" in l]
    assert len(mf) == 108, len(mf)
    assert all("cbsz:2 blgp:2" in body[i] for i in mf)
    ops = [re.search(r"f8f6f4 (a\[\d+:\d+\]), ([av]\[(\d+):(\d+)\]), ([av]\[(\d+):(\d+)\])", body[i]) for i in mf]
    assert all(o and int(o.group(4)) - int(o.group(3)) == 5 and int(o.group(7)) - int(o.group(6)) == 5 for o in ops)
    assert len({o.group(1) for o in ops[:72]}) == 6
    # its fragment reads: per k-step one visible ds_read_b128 and one asm ds_read_b64 (24 of the slot's 32 bytes); nothing may touch an
    # asm read's registers before an lgkmcnt wait (hipcc does not know the read exists)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mfma_hazard_scan as hz
    for kern in ("knn_score6_kernel", "knn_probe6_kernel"):
        n_asm, unwaited = hz.asm_lds_reads_are_waited_for(str(out), kern)
        assert n_asm == 36 and unwaited == 0, (kern, n_asm, unwaited)
    meta = "\n".join(ls)
    m6 = re.search(r"\.name:\s+\S*knn_score6_kernel\S*\n(.*?)\.wavefront_size", meta, re.S).group(1)
    assert re.search(r"\.vgpr_spill_count:\s+0\b", m6) and re.search(r"\.private_segment_fixed_size:\s+0\b", m6), m6


def test_network_constructors_keep_the_reference_signature():
    """content_encoder.py:9-14 / f0_estimator.py:9-14: the reference's constructors take sizes.  The defaults (the only ones the three
    scripts construct) run through the fused entry points; other sizes build the reference's state_dict for those sizes and run layer
    by layer through the op-level C ABI (module/_generic.py; numerics: tests/test_gpu_models.py)"""
    from module import schema
    from module.content_encoder import ContentEncoder
    from module.f0_estimator import F0Estimator
    ce = ContentEncoder(n_fft=1280, internal_channels=512, hidden_channels=1536, output_channels=768, num_layers=4)
    pe = F0Estimator(n_fft=1280, internal_channels=256, hidden_channels=512, output_channels=4096, num_layers=4)
    assert not ce.generic and not pe.generic
    ce2 = ContentEncoder(hidden_channels=1024, num_layers=2)
    assert ce2.generic and ce2.state_dict()["mid_layers.1.pw_conv1.weight"].shape == (1024, 512, 1)
    assert "mid_layers.2.scale" not in ce2.state_dict()
    pe2 = F0Estimator(output_channels=2048)
    assert pe2.generic and pe2.state_dict()["output_layer.weight"].shape == (2048, 256, 1)
    assert set(pe.state_dict()) == set(schema.f0_estimator_schema())
    with pytest.raises(RuntimeError):
        pe2.forward(torch.zeros(1, 641, 8))              # no CPU path
