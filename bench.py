"""bench.py -- voice-conversion throughput of the MI355X hot path (contract: see the task prompt).

One "step" = one pass of the whole hot path over one batch of synthetic input that is already
resident in HBM: `--utterances` x `--seconds` of 16 kHz audio cut into the reference's overlapping
windows (inference.py:94-101) -> magnitude STFT -> F0 estimator (+ per-window pitch transform) ->
content encoder -> kNN match against an M-vector voice library -> decoder -> waveform windows.
Default workload = BASELINE.json config 3/4 on one GPU: 64 utterances x 10 s (384 windows,
172 800 content frames per step) against a 1 M-vector library.

N > 1: `python bench.py --gpus N` launches its own N ranks (torch.distributed.run, one process per GPU,
before this process has touched a GPU) unless it already runs under a launcher (RANK in the
environment).  Windows are independent units, so the headline ranks are pure data-parallel replicas over
disjoint utterance sets ("weak" scaling, no data-path collective).  BASELINE config 4 -- the library cut
into N row slabs, features all-gathered, per-shard exact top-k all-gathered and merged -- runs as a
second leg on one fixed global batch, is checked bitwise against the replicated path on every rank and
is reported as `sharded_knn`, never as `value`.

Prints ONE JSON line on rank 0.  Secondary legs (`--legs`) sit beside the headline in the same line.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "alive-vc_amd"))

import torch                                              # noqa: E402
import torch.distributed as dist                           # noqa: E402

FRAME = 320
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_FP8_TFLOPS = 5000.0           # dense fp8 MFMA peak (block-scaled 32x32x64 e4m3), same table
PEAK_FP6_TFLOPS = 10000.0          # dense fp6 / fp4 MFMA peak (block-scaled 32x32x64 e2m3: "FP6 at FP4 rate"), same table
ALL_LEGS = ("uncorrelated", "fp8_prefilter", "decoder_split_bf16", "encoder_bf16x6", "bf16_prefilter", "strict_knn", "reference_grade", "clustered_library", "overlap_shared", "context_trim", "cli_default", "library_build",
            "library_build_1m", "pcie_inclusive", "e2e_24k", "config2", "streaming", "cpu_baseline")


# --------------------------------------------------------------------------------------- launch
def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (one process per GPU) and pass rank 0's
    JSON line through.  Nothing in this process has initialised a GPU (torch.cuda.device_count() does not), and the
    children are separate processes, not an exec of this one."""
    have = torch.cuda.device_count()
    if not args.same_device and have < args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but this node shows {have} GPU(s)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # --standalone: the launcher picks (and holds) a free rendezvous port itself -- binding a socket here, closing it and
    # passing the number on would race with every other job on the node
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    # stdout of this process is the contract's ONE JSON line: everything else the ranks (or their libraries: gloo announces
    # its peers on stdout) print goes to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        ok = line.startswith("{") and line.rstrip().endswith("}")
        (sys.stdout if ok else sys.stderr).write(line)
        (sys.stdout if ok else sys.stderr).flush()
    return proc.wait()


# --------------------------------------------------------------------------------------- inputs
def synth_signals(n, L, device, seed, f_lo=90.0, f_step=7.0, f_list=None):
    """n independent synthetic signals [n, L] at 16 kHz: noise + three partials, peak-normalised like inference.py:92"""
    g = torch.Generator(device=device).manual_seed(seed)
    t = torch.arange(L, device=device, dtype=torch.float32) / 16000.0
    out = torch.empty(n, L, device=device)
    for u in range(n):
        f = f_list[u] if f_list is not None else f_lo + f_step * u
        x = 0.1 * torch.randn(L, device=device, generator=g)
        for h in (1, 2, 3):
            x = x + 0.2 / h * torch.sin(2 * torch.pi * f * h * t)
        out[u] = x / x.abs().max()
    return out


def synth_windows(n_utt, seconds, chunk, device, seed):
    """n_utt synthetic utterances -> all of their overlapping windows (inference.py:94-101)."""
    from module.pipeline import make_windows
    sig = synth_signals(n_utt, int(seconds * 16000), device, seed)
    return torch.cat([make_windows(sig[u:u + 1], chunk)[0] for u in range(n_utt)], 0).contiguous()


def ce_derived_tokens(conv, M, device, seed=4321, L=144000):
    """A DENSE library: M content-encoder frames of synthetic audio from the same signal family as the bench batch
    ("also run kNN with CE-derived vectors") -- the frames of a query then have many library rows at
    nearly the same cosine, which is what a real single-speaker library looks like to the candidate stage."""
    from module.spectrogram import spectrogram
    lf = L // FRAME
    n = (M + lf - 1) // lf
    g = torch.Generator(device="cpu").manual_seed(seed)
    freqs = (80.0 + 420.0 * torch.rand(n, generator=g)).tolist()
    toks = torch.empty(768, n * lf, device=device)
    for i in range(0, n, 128):
        sig = synth_signals(min(128, n - i), L, device, seed + 1 + i, f_list=freqs[i:i + 128])
        feat = conv.ce(spectrogram(sig))                                         # [b, 768, lf]
        toks[:, i * lf:(i + sig.shape[0]) * lf] = feat.permute(1, 0, 2).reshape(768, -1)
    return toks[:, :M].contiguous()


def usable_cores():
    """host cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box
    shows 256 logical CPUs under a 16-CPU quota; 256 threads there run the oracle 25x slower)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(windows, rows_dev, args):
    """The CPU oracle (oracle/alive_oracle.py, proven equal to the reference in the build container) on a
    bounded sample of the same workload, all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import alive_oracle as O
    from module import schema, synthetic
    cores = usable_cores()
    torch.set_num_threads(cores)
    ce = synthetic.make_state_dict(schema.content_encoder_schema(), 2, "ce.")
    pe = synthetic.make_state_dict(schema.f0_estimator_schema(), 2, "pe.")
    dec = synthetic.make_state_dict(schema.decoder_schema(), 2, "dec.")
    lib = rows_dev.t().contiguous().cpu().unsqueeze(0)          # [1, 768, M] fp32 on the host
    done, t_total, budget = 0, 0.0, args.cpu_seconds
    with torch.no_grad():
        while done < windows.shape[0]:
            w = windows[done:done + 1].cpu()
            t0 = time.perf_counter()
            O.convert_window(ce, pe, dec, w, lib, k=args.k, alpha=0.0)
            t_total += time.perf_counter() - t0
            done += 1
            if t_total + t_total / done > budget:
                break
    frames = done * (windows.shape[1] // FRAME)
    return {"value": round(frames / t_total, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{done} window(s) x {windows.shape[1] // FRAME} frames of the same batch, same {lib.shape[2]}-vector "
                      f"library, PyTorch-CPU oracle, {cores} threads, {t_total:.1f} s"}


class ScoreTimer:
    """per-launch HIP-event timing of the dominant kernel (the MFMA candidate scoring) on the stream it is launched on"""

    def __init__(self, nat):
        self.nat, self.pairs, self.whole = nat, [], []      # kernel events; events around the whole tiered search

    def wrap(self, library):
        orig = library.search

        def timed(src, k):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()                               # materialise the handles
            self.pairs.append((a, b, src.shape[0] * src.shape[2], library.M))
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.whole.append((s0, s1))
            s0.record()
            try:
                return orig(src, k, events=(a, b))               # recorded by the library around the scoring kernel, on its stream
            finally:
                s1.record()
        library.search = timed
        return orig

    @staticmethod
    def unwrap(library):
        library.__dict__.pop("search", None)

    def clear(self):
        self.pairs.clear()
        self.whole.clear()

    def search_ms(self):
        """whole tiered search (all tiers, rescoring included), per call (after a synchronize)"""
        return sum(a.elapsed_time(b) for a, b in self.whole) / max(1, len(self.whole))

    def totals(self):
        """(ms, flop, launches) of everything recorded since the last clear (call after a synchronize)"""
        ms = sum(a.elapsed_time(b) for a, b, _, _ in self.pairs)
        flop = sum(2.0 * 768 * m * t for _, _, t, m in self.pairs)
        return ms, flop, len(self.pairs)


def guarded(fn):
    """a secondary leg must never cost the headline line"""
    try:
        return fn()
    except Exception as e:                                    # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def timed_steps(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--library-size", type=int, default=1_000_000)
    ap.add_argument("--utterances", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--chunk", type=int, default=48000)
    ap.add_argument("-k", type=int, default=4)
    ap.add_argument("--window-batch", type=int, default=192)      # (384 windows = 2 batches; 128 / 192 / 256 / 384: 188.4 / 187.4 / 187.4 / 188.5 ms per step, tools/sweep_streams.sh)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU-oracle leg (0 = skip)")
    ap.add_argument("--legs", default="all", help="comma list of secondary legs (N = 1): " + ",".join(ALL_LEGS) + " | all | none")
    ap.add_argument("--no-nets-roofline", action="store_true", help="skip the separate timed passes of the networks (profiling runs: keeps the kernel trace to the steps)")
    ap.add_argument("--stream-steps", type=int, default=1000, help="steps of the streaming leg (BASELINE config 5)")
    ap.add_argument("--no-shard-library", action="store_true", help="skip the library-sharded leg (config 4) of a multi-GPU run")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo + --same-device only exercise the multi-rank code path on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true", help="test only: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 only: create a world-size-1 process group anyway and run the library-sharded leg with one shard, so "
                         "that RCCL init, the device-tensor collectives and the fences of module/sharded.py execute on a 1-GPU box")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args))
    if args.force_dist and "RANK" not in os.environ:              # a one-rank "job" without a launcher: rendezvous on loopback
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))

    # stdout of a rank carries the contract's ONE JSON line and nothing else: libraries that print to file descriptor 1
    # themselves (RCCL's version banner at communicator init, gloo's peer announcements) are pointed at stderr, and the line
    # goes to a private duplicate of the original descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", 0 if args.same_device else local)
    torch.cuda.set_device(dev)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    legs = set(ALL_LEGS) if args.legs == "all" else set(x for x in args.legs.split(",") if x and x != "none")
    if args.cpu_seconds <= 0:
        legs.discard("cpu_baseline")
    if rank != 0 or world > 1:
        legs = set()

    from module import _native as nat
    from module.common import PackedLibrary
    from module.content_encoder import ContentEncoder
    from module.decoder import Decoder
    from module.f0_estimator import F0Estimator
    from module.pipeline import Converter
    nat.lib()                                                # fail loudly if the HIP library is missing

    # ---- resident state: weights, library, input windows ----
    M = args.library_size
    g = torch.Generator(device=dev).manual_seed(1234)         # same library on every rank
    tokens = torch.randn(768, M, device=dev, generator=g)
    library = PackedLibrary(tokens)
    if not dist_on or args.no_shard_library:
        del tokens
    conv = Converter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), dev).set_library(library)
    windows = synth_windows(args.utterances, args.seconds, args.chunk, dev, seed=100 + rank)
    n_win, L = windows.shape
    frames_per_step = n_win * (L // FRAME)
    useful_frames = args.utterances * int(args.seconds * 16000) // FRAME

    timer = ScoreTimer(nat)
    timer.wrap(library)

    def step():
        return conv.convert_windows(windows, k=args.k, window_batch=args.window_batch)

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    timer.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    assert torch.isfinite(out).all(), "non-finite waveform"
    stats = library.search_stats()                            # of the last timed step's search

    # ---- roofline of the scoring kernel ----
    ms, flops, launches = timer.totals()
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    stage = library.prefilter                                 # "fp6" (default since round 5), "fp8" or "bf16"
    fp8 = stage in ("fp8", "fp6")
    peak = {"fp6": PEAK_FP6_TFLOPS, "fp8": PEAK_FP8_TFLOPS}.get(stage, PEAK_BF16_TFLOPS)
    # `traffic` (HBM bytes per launch from PMC counters) cannot be collected inside this run -- counters need rocprofv3 around the
    # process -- so it is null here; what the committed profile of the same kernel on the same shape measured is reported beside it
    # under a name of its own, with the file it comes from
    traffic_profiled = None
    pmc_name = {"fp6": "knn_score6_pmc.json", "fp8": "knn_score8_pmc.json"}.get(stage, "knn_score_pmc.json")
    pmc = os.path.join(ROOT, "profiles", pmc_name)
    if os.path.exists(pmc):
        pj = json.load(open(pmc))
        traffic_profiled = {"hbm_bytes_per_launch": pj.get("hbm_bytes_per_launch"), "algorithmic_bytes_per_launch": pj.get("algorithmic_bytes_per_launch"),
                            "mfma_pipe_utilisation": pj.get("mfma_pipe_utilisation"), "l2_hit_rate": pj.get("l2_hit_rate"),
                            "source_commit": pj.get("source_commit"),
                            "source": "profiles/" + pmc_name + " (rocprofv3 --pmc passes of tools/pmc_knn8.sh over the same kernel and shape, not this run)"}
    roofline = {"kernel": {"fp6": "knn_score6_kernel", "fp8": "knn_score8_kernel"}.get(stage, "knn_score_kernel"), "bound": "mfma",
                "achieved": round(achieved, 1),
                "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None, "traffic_profiled": traffic_profiled,
                "mfma_dtype": {"fp6": "fp6 e2m3 (both operands), block-scaled 32x32x64", "fp8": "fp8 e4m3, block-scaled 32x32x64"}.get(stage, "bf16 32x32x16"),
                "frac_of_fp8_peak": round(achieved / PEAK_FP8_TFLOPS, 4) if stage == "fp6" else None,
                "launches": launches, "avg_launch_ms": round(ms / max(1, launches), 3),
                "kernel_share_of_step": round(ms * 1e-3 / dt, 3), "search_ms": round(timer.search_ms(), 2),
                "search_tiers_last_step": stats}

    # ---- roofline of the networks (the other 60 % of the step), measured live: the front end (spectrogram, f0 estimator, content
    #      encoder) and the decoder once more over the same window batches, each family alone on the current stream between two
    #      events.  Algorithmic FLOP per frame from SURVEY.md 8(d): DFT 3.3 + PE 4.54 + CE 14.05 MFLOP (front end, six bf16 MFMA
    #      products per fp32-grade product: effective peak 2.5 PF / 6) and fe 18.12 + ho 0.07 + filter 78.40 MFLOP (decoder, three
    #      products per product: 2.5 PF / 3).  The step-level fraction prices the whole step against those three peaks.
    def nets_roofline():
        wb = args.window_batch
        feat = torch.empty(n_win, 768, L // FRAME, device=dev)
        f0 = torch.empty(n_win, 1, L // FRAME, device=dev)
        wav = torch.empty(n_win, L, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        def run():
            ev[0].record()
            for i in range(0, n_win, wb):
                conv.features(windows[i:i + wb], out=(feat[i:i + wb], f0[i:i + wb]))
            ev[1].record()
            for i in range(0, n_win, wb):
                conv.dec(feat[i:i + wb], f0[i:i + wb], out=wav[i:i + wb])
            ev[2].record()
            torch.cuda.synchronize()
            return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
        run()
        t = [run() for _ in range(2)]
        ms_enc, ms_dec = min(x[0] for x in t), min(x[1] for x in t)
        fl_enc = (3.3 + 4.54 + 14.05) * 1e6 * frames_per_step
        fl_dec = (18.12 + 0.07 + 78.40) * 1e6 * frames_per_step
        # decoder precision mode 1 (default since round 5): the four ConvNeXt layers' pointwise convs (4 x 2 x 2*512*1536 = 12.58 MFLOP per
        # frame) and the six k = 5 convs of the 256-channel FilterBlock (6 x 2*256*256*5 x 10 columns = 39.32) run ONE bf16 MFMA per
        # product, and so do the norm-FiLM projection (4.19), the two coarse down convs (2.62 + 1.31) and the mid conv (0.66): 60.68 MFLOP
        # priced at 2.5 PF; the other 35.91 MFLOP run three.  The family's peak is the blend: its FLOP / its ideal time.
        from module import ops as _ops
        dec_mode = 2 if conv.dec._split_for_this_checkpoint() else _ops.decoder_precision(0)
        # ... and the six k5 convs of the fused 64-channel FilterBlock (6 x 2*64*64*5 x 80 columns = 19.66)
        fl_dec_plain = (12.58 + 39.32 + 4.19 + 2.62 + 1.31 + 0.66 + 19.66) * 1e6 * frames_per_step if dec_mode == 1 else 0.0
        dec_ideal_s = fl_dec_plain / (PEAK_BF16_TFLOPS * 1e12) + (fl_dec - fl_dec_plain) / (PEAK_BF16_TFLOPS / 3 * 1e12)
        # encoder precision mode 1 (default since round 5): the ConvNeXt pointwise convs (CE 12.58 + PE 2.10 MFLOP per frame) run three fp16
        # MFMAs per product, the rest of the front end (DFT, input / output layers, classifier: 7.21 MFLOP) six bf16 MFMAs
        enc_mode = _ops.encoder_precision(0)
        fl_enc_3 = (12.58 + 2.10) * 1e6 * frames_per_step if enc_mode == 1 else 0.0
        enc_ideal_s = fl_enc_3 / (PEAK_BF16_TFLOPS / 3 * 1e12) + (fl_enc - fl_enc_3) / (PEAK_BF16_TFLOPS / 6 * 1e12)
        pk_enc, pk_dec = fl_enc / enc_ideal_s / 1e12, fl_dec / dec_ideal_s / 1e12
        fl_knn = 2.0 * 768 * M * frames_per_step
        ideal_ms = (fl_knn / (peak * 1e12) + dec_ideal_s + enc_ideal_s) * 1e3
        fam = lambda ms_, fl, pk, what: {"what": what, "ms_per_step": round(ms_, 2), "algorithmic_tflop": round(fl / 1e12, 2),
                                         "achieved": round(fl / (ms_ * 1e-3) / 1e12, 1), "peak": round(pk, 1), "unit": "TFLOP/s",
                                         "frac": round(fl / (ms_ * 1e-3) / 1e12 / pk, 4)}
        return {"bound": "mfma", "timing": "HIP events on the launching stream, each family alone, one stream (the step overlaps window batches on side streams)",
                "counters": "profiles/nets_pmc.json (per kernel: MFMA-pipe utilisation, bytes beyond L2, LDS conflicts, VALU co-execution; tools/pmc_nets.sh)",
                "front_end": dict(fam(ms_enc, fl_enc, pk_enc, ("spectrogram (DFT as a GEMM) + F0Estimator + ContentEncoder: fp16 split planes (3 MFMAs per product, fp32-grade) for the ConvNeXt pointwise convs "
                                                                "(14.7 of the 21.9 MFLOP per frame), 3-plane split bf16 (6 MFMAs per product) for the DFT, the input / output layers and the classifier; peak = the blend" if enc_mode == 1 else
                                                                "spectrogram (DFT as a GEMM) + F0Estimator + ContentEncoder: 3-plane split bf16, 6 MFMAs per product")),
                                  precision_mode=enc_mode, frac_of_bf16x6_peak=round(fl_enc / (ms_enc * 1e-3) / 1e12 / (PEAK_BF16_TFLOPS / 6), 4)),
                "decoder": dict(fam(ms_dec, fl_dec, pk_dec, ("FeatureExtractor + HarmonicOscillator + Filter: plain fp16 (1 MFMA per product) for the ConvNeXt pointwise convs, the k5 convs of the 256- and 64-channel FilterBlocks, the norm-FiLM projection, the two coarse down convs and the mid conv "
                                                              "(80.3 of the 96.6 MFLOP per frame), 2-plane split bf16 (3 MFMAs per product) for the rest; peak = the blend, FLOP / ideal time" if dec_mode == 1 else
                                                              "FeatureExtractor + HarmonicOscillator + Filter: 2-plane split bf16, 3 MFMAs per product") +
                                                             " (every FilterBlock one fused kernel: 256 / 64 channels csrc/filter_big.hip, 16 / 8 channels csrc/filter_small.hip; exact f32 MFMA only for the strided / transposed convs of the two finest scales)"),
                                precision_mode=dec_mode, frac_of_split_bf16_peak=round(fl_dec / (ms_dec * 1e-3) / 1e12 / (PEAK_BF16_TFLOPS / 3), 4)),
                "step": {"ideal_ms": round(ideal_ms, 1), "ms_per_step": round(dt / args.steps * 1e3, 2), "frac": round(ideal_ms / (dt / args.steps * 1e3), 4),
                         "ideal": "kNN 2*768*M*T FLOP at the candidate stage's MFMA peak + decoder at its blended peak (2.5 PF for the plain-fp16 layers, 2.5 PF / 3 for the rest) + front end at its blended peak (2.5 PF / 3 for the fp16-split layers, 2.5 PF / 6 for the rest)"}}
    roofline_nets = guarded(nets_roofline) if rank == 0 and not args.no_nets_roofline else None

    extra = {}
    # Data dependence of the scoring kernel: the candidate-list path is taken more often when the frames of a wave
    # are uncorrelated.  The batch above comes from one synthetic signal family (correlated frames); time the same
    # kernel once on i.i.d. Gaussian frames so both ends are on record (real speech lies in between).
    if "uncorrelated" in legs:
        def leg():
            timer.clear()
            rnd = torch.randn(n_win, 768, L // FRAME, device=dev, generator=torch.Generator(device=dev).manual_seed(77))
            library.search(rnd, args.k)
            torch.cuda.synchronize()
            ms_u, fl_u, _ = timer.totals()
            roofline["achieved_uncorrelated_frames"] = round(fl_u / (ms_u * 1e-3) / 1e12, 1)
            roofline["search_tiers_uncorrelated"] = library.search_stats()
            return None
        err = guarded(leg)
        if err:
            roofline["uncorrelated_error"] = err
    timer.unwrap(library)

    # The same step with the fp8 candidate stage, the default of rounds 2-4 (ALIVE_KNN_PREFILTER=fp8): same library object
    if "fp8_prefilter" in legs and library.prefilter != "fp8":
        def leg():
            lib8 = library.with_prefilter("fp8")
            t8 = ScoreTimer(nat)
            t8.wrap(lib8)
            conv.set_library(lib8)
            try:
                conv.convert_windows(windows, k=args.k, window_batch=args.window_batch)
                t8.clear()
                tb, o8 = timed_steps(step, 2)
                ms8, fl8, n8 = t8.totals()
                return {"ms_per_step": round(tb * 1e3, 2), "frames_per_s": round(frames_per_step / tb, 1),
                        "scoring_tflops": round(fl8 / (ms8 * 1e-3) / 1e12, 1), "scoring_ms_per_launch": round(ms8 / n8, 3),
                        "frac_of_fp8_peak": round(fl8 / (ms8 * 1e-3) / 1e12 / PEAK_FP8_TFLOPS, 4), "search_ms": round(t8.search_ms(), 2),
                        "waveforms_equal_default_stage": bool(torch.equal(o8, out)), "search_tiers": lib8.search_stats()}
            finally:
                conv.set_library(library)
        extra["fp8_prefilter"] = guarded(leg)

    # The same step with the decoder's plain-fp16 layers back on two-plane split bf16 (alive_decoder_precision 2, the arithmetic of rounds
    # 1 - 4): what the default mode buys, and how far its waveforms are from the split form's
    if "decoder_split_bf16" in legs:
        def leg():
            from module import ops as _ops
            mode0 = _ops.decoder_precision(0)
            if mode0 != 1:
                return {"skipped": "the headline already runs decoder precision mode %d" % mode0}
            _ops.decoder_precision(2)
            try:
                conv.convert_windows(windows, k=args.k, window_batch=args.window_batch)
                t2, o2 = timed_steps(step, 2)
            finally:
                _ops.decoder_precision(mode0)
            d = (o2.double() - out.double())
            return {"ms_per_step": round(t2 * 1e3, 2), "frames_per_s": round(frames_per_step / t2, 1),
                    "waveform_rms": round(float(o2.double().pow(2).mean().sqrt()), 5),
                    "headline_minus_this_rms": float("%.3e" % d.pow(2).mean().sqrt().item()), "headline_minus_this_max": float("%.3e" % d.abs().max().item()),
                    "note": "ALIVE_DECODER_PRECISION=2: every decoder GEMM on two-plane split bf16 (3 MFMAs per product); the headline runs the "
                            "ConvNeXt pointwise convs, the 256-channel FilterBlock's k5 convs and four smaller layers, and the 64-channel FilterBlock's k5 convs, on plain fp16"}
        extra["decoder_split_bf16"] = guarded(leg)

    # The same step with the encoders' ConvNeXt pointwise convs back on three bf16 planes, six MFMAs per product (alive_encoder_precision 2,
    # the arithmetic of rounds 1 - 4); the headline runs them on fp16 split planes, three MFMAs per product, both fp32-grade
    if "encoder_bf16x6" in legs:
        def leg():
            from module import ops as _ops
            mode0 = _ops.encoder_precision(0)
            if mode0 != 1:
                return {"skipped": "the headline already runs encoder precision mode %d" % mode0}
            _ops.encoder_precision(2)
            try:
                conv.convert_windows(windows, k=args.k, window_batch=args.window_batch)
                t2, o2 = timed_steps(step, 2)
            finally:
                _ops.encoder_precision(mode0)
            d = (o2.double() - out.double())
            return {"ms_per_step": round(t2 * 1e3, 2), "frames_per_s": round(frames_per_step / t2, 1),
                    "headline_minus_this_rms": float("%.3e" % d.pow(2).mean().sqrt().item()),
                    "windows_bitwise_equal": int((d.abs().amax(dim=-1) == 0).sum().item()), "windows": int(d.shape[0]),
                    "note": "ALIVE_ENCODER_PRECISION=2: the ConvNeXt pointwise convs of ContentEncoder / F0Estimator on three bf16 planes (6 MFMAs per "
                            "product); the headline runs them on fp16 split planes (3 MFMAs per product).  Both reproduce the reference's content "
                            "features to 7.7e-7 of their RMS and all 450 f0 classes of the fixture (tools/enc_precision_check.py); a window differs "
                            "where a feature ulp moves a near-tied neighbour or an f0 class"}
        extra["encoder_bf16x6"] = guarded(leg)

    # The same step with the bf16 candidate stage (ALIVE_KNN_PREFILTER=bf16): same library object, fp8 image unused
    if "bf16_prefilter" in legs:
        def leg():
            lib16 = library.with_prefilter("bf16")
            t16 = ScoreTimer(nat)
            t16.wrap(lib16)
            conv.set_library(lib16)
            try:
                conv.convert_windows(windows, k=args.k, window_batch=args.window_batch)
                t16.clear()
                tb, o16 = timed_steps(step, 2)
                ms16, fl16, n16 = t16.totals()
                return {"ms_per_step": round(tb * 1e3, 2), "frames_per_s": round(frames_per_step / tb, 1),
                        "scoring_tflops": round(fl16 / (ms16 * 1e-3) / 1e12, 1), "scoring_ms_per_launch": round(ms16 / n16, 3),
                        "frac_of_bf16_peak": round(fl16 / (ms16 * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                        "search_ms": round(t16.search_ms(), 2),
                        "waveforms_equal_default_stage": bool(torch.equal(o16, out)), "search_tiers": lib16.search_stats()}
            finally:
                conv.set_library(library)
        extra["bf16_prefilter"] = guarded(leg)

    # The same step with the STRICT search (ALIVE_KNN_STRICT=1): bf16 candidate stage under a deterministic Cauchy-Schwarz bound,
    # no statistical assumption anywhere -- the price of a guarantee, and a check that it returns the very same waveforms
    if "strict_knn" in legs:
        def leg():
            libs = library.with_strict()
            ts_ = ScoreTimer(nat)
            ts_.wrap(libs)
            conv.set_library(libs)
            try:
                conv.convert_windows(windows, k=args.k, window_batch=args.window_batch)
                ts_.clear()
                tb, os_ = timed_steps(step, 2)
                return {"ms_per_step": round(tb * 1e3, 2), "frames_per_s": round(frames_per_step / tb, 1),
                        "search_ms": round(ts_.search_ms(), 2), "waveforms_equal_default_search": bool(torch.equal(os_, out)),
                        "search_tiers": libs.search_stats(),
                        "certificate": "deterministic: |stage - exact| <= ||q - bf16(q)|| + 1.004 max_R ||r - bf16(r)|| + 1e-4"}
            finally:
                conv.set_library(library)
        extra["strict_knn"] = guarded(leg)

    # ONE timed loop with everything at reference-grade arithmetic: encoders on three bf16 planes (mode 2), every
    # decoder GEMM on two-plane split bf16 (mode 2) and the STRICT search (bf16 candidates under the deterministic certificate: bit-exact
    # top-k without an "unless") -- what the step costs when none of this round's and last round's format levers is pulled
    if "reference_grade" in legs:
        def leg():
            from module import ops as _ops
            e0, d0 = _ops.encoder_precision(0), _ops.decoder_precision(0)
            libs = library.with_strict()
            conv.set_library(libs)
            try:
                _ops.encoder_precision(2)
                _ops.decoder_precision(2)
                conv.convert_windows(windows, k=args.k, window_batch=args.window_batch)
                tb, o_ = timed_steps(step, 2)
                st_ = libs.search_stats()
            finally:
                _ops.encoder_precision(e0)
                _ops.decoder_precision(d0)
                conv.set_library(library)
            d = (o_.double() - out.double())
            return {"ms_per_step": round(tb * 1e3, 2), "frames_per_s": round(frames_per_step / tb, 1),
                    "headline_minus_this_rms": float("%.3e" % d.pow(2).mean().sqrt().item()), "headline_minus_this_max": float("%.3e" % d.abs().max().item()),
                    "waveform_rms": round(float(o_.double().pow(2).mean().sqrt()), 5),
                    "windows_bitwise_equal_headline": int((d.abs().amax(dim=-1) == 0).sum().item()), "windows": int(d.shape[0]),
                    "search_tiers": st_,
                    "arithmetic": "ALIVE_ENCODER_PRECISION=2 (3 bf16 planes, 6 MFMAs per product) + ALIVE_DECODER_PRECISION=2 (2 bf16 planes, 3 MFMAs) "
                                  "+ ALIVE_KNN_STRICT=1 (bf16 MFMA candidates, deterministic certificate, exact fp32 rescoring), one timed loop"}
        extra["reference_grade"] = guarded(leg)

    # A dense, CE-derived 1 M-vector library (every query frame has many rows at nearly its best cosine): the case in
    # which a candidate stage on fp8 cannot be certified and the search has to fall through its tiers.
    if "clustered_library" in legs:
        def leg():
            toks = ce_derived_tokens(conv, M, dev)
            res = {}
            outs = {}
            for pf in (library.prefilter, "bf16") if library.prefilter != "bf16" else ("fp8", "bf16"):
                libc = PackedLibrary(toks, prefilter=pf)
                tc = ScoreTimer(nat)
                tc.wrap(libc)
                conv.set_library(libc)
                conv.convert_windows(windows, k=args.k, window_batch=args.window_batch)
                tc.clear()
                tb, outs[pf] = timed_steps(step, 2)
                msc, flc, nc = tc.totals()
                res[pf] = {"ms_per_step": round(tb * 1e3, 2), "frames_per_s": round(frames_per_step / tb, 1),
                           "search_ms": round(tc.search_ms(), 2),
                           "search_effective_tflops": round(flc / nc / (tc.search_ms() * 1e-3) / 1e12, 1),
                           "search_tiers": libc.search_stats()}
                del libc
            first = [pf for pf in res if pf != "bf16"][0]
            res["default_vs_bf16_prefilter"] = round(res[first]["ms_per_step"] / res["bf16"]["ms_per_step"], 4)
            res["waveforms_equal_across_stages"] = bool(torch.equal(outs[first], outs["bf16"]))
            res["library"] = f"{M} content-encoder frames of synthetic audio of the bench batch's signal family (dense: SURVEY 8(d))"
            return res
        try:
            extra["clustered_library"] = guarded(leg)
        finally:
            conv.set_library(library)

    # Optional mode of the build, reported beside the headline and never as `value`: the windows of an utterance overlap by two
    # thirds, and a content / f0 frame away from its window's edges has the same value in every window that holds it -- so
    # the front end (spectrogram, f0 estimator, content encoder, kNN match) can run once per utterance instead of once per
    # window.  Every waveform sample is bitwise that of the headline step (checked here on the whole batch).
    if "overlap_shared" in legs:
        def leg():
            per = n_win // args.utterances
            sh_step = lambda: conv.convert_windows(windows, k=args.k, window_batch=args.window_batch, share_overlap=per)  # noqa: E731
            sh_step()
            ts, out_s = timed_steps(sh_step, 2)
            return {"ms_per_step": round(ts * 1e3, 2), "frames_per_s": round(frames_per_step / ts, 1),
                    "useful_frames_per_s": round(useful_frames / ts, 1), "rtf": round(ts / (args.utterances * args.seconds), 6),
                    "waveforms_bitwise_equal_headline": bool(torch.equal(out_s, out)),
                    "frames_through_front_end_and_match": conv.last_front_end_frames, "of": frames_per_step,
                    "note": "front end once per utterance + the two 14-frame edge blocks of every window; pitch transform and decoder per window"}
        extra["overlap_shared"] = guarded(leg)

    # Optional mode of the build, reported beside the headline and never as `value`: inference.py keeps the centre third of
    # every window, so only the frames that can reach it through the decoder need the kNN match (Converter(keep_frames=...),
    # `--trim-context`).  The kept samples are bitwise those of the full computation (checked here on the whole batch).
    if "context_trim" in legs:
        def leg():
            cf = (L // FRAME) // 3
            ref_out = out[:, cf * FRAME:2 * cf * FRAME].clone()
            trim_step = lambda: conv.convert_windows(windows, k=args.k, window_batch=args.window_batch, keep_frames=(cf, 2 * cf))  # noqa: E731
            trim_step()                                           # warm-up (scratch sizes)
            tt0, out_t = timed_steps(trim_step, 2)
            same = bool(torch.equal(out_t[:, cf * FRAME:2 * cf * FRAME], ref_out))
            return {"ms_per_step": round(tt0 * 1e3, 2), "windows_per_s": round(n_win / tt0, 1),
                    "useful_frames_per_s": round(useful_frames / tt0, 1), "kept_samples_bitwise_equal": same,
                    "note": "kNN match and decoder on frames [cf-32, 2cf+16) of each window (oscillator phase over the whole window), content encoder on that range +-16; spectrogram and f0 estimator on the whole window"}
        extra["context_trim"] = guarded(leg)

    # What inference.py does by default since round 5 (never `value`): overlap sharing AND context trimming together -- the front
    # end once per utterance, the kNN match on the union of the windows' trimmed ranges, the decoder on frames [cf-32, 2cf+16)
    # of every window.  The kept centre thirds (all the CLI writes) are bitwise those of the headline step.
    if "cli_default" in legs:
        def leg():
            cf = (L // FRAME) // 3
            per = n_win // args.utterances
            ref_out = out[:, cf * FRAME:2 * cf * FRAME].clone()
            cli_step = lambda: conv.convert_windows(windows, k=args.k, window_batch=args.window_batch, share_overlap=per,  # noqa: E731
                                                    keep_frames=(cf, 2 * cf))
            cli_step()
            tc, out_c = timed_steps(cli_step, 2)
            return {"ms_per_step": round(tc * 1e3, 2), "useful_frames_per_s": round(useful_frames / tc, 1),
                    "rtf": round(tc / (args.utterances * args.seconds), 6),
                    "kept_samples_bitwise_equal_headline": bool(torch.equal(out_c[:, cf * FRAME:2 * cf * FRAME], ref_out)),
                    "frames_through_match": conv.last_front_end_frames, "of": frames_per_step,
                    "note": "Converter.convert(share_overlap='auto', trim_context=True), the CLI's default: spectrogram + f0 estimator + content "
                            "encoder once per utterance (+ the f0 estimator on the two 14-frame edge blocks of every window), kNN match on "
                            "frames [cf-32, (per-1) cf + 2cf+16) of every utterance, decoder on frames [cf-32, 2cf+16) of every window"}
        extra["cli_default"] = guarded(leg)

    # SURVEY 8 f3 at scale (never `value`): generate_voice_library.py's device work for a 200 000-vector bank -- the content encoder
    # over a synthetic corpus of 25 000 clips of 7 680 samples (the reference's clip length: 24 frames each), eight frames per clip
    # (the reference draws from frames 0..7), then --dedup 0.98 (the library's own kNN kernel against itself + alive_dedup_pass)
    def library_build_leg(n_clips, noise=0.05):
        import generate_voice_library as G
        fpc = 8
        gl = torch.Generator(device=dev).manual_seed(77)
        t = torch.arange(G.CLIP, device=dev, dtype=torch.float32) / 16000.0
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        mem0 = torch.cuda.memory_allocated(dev)
        toks = torch.empty(768, n_clips * fpc, device=dev)
        t0 = time.perf_counter()
        for s0 in range(0, n_clips, 1000):
            nb = min(1000, n_clips - s0)
            f = 80.0 + 400.0 * torch.rand(nb, 1, device=dev, generator=gl)
            clips = 0.4 * torch.sin(2 * math.pi * f * t) + 0.2 * torch.sin(2 * math.pi * 2.7 * f * t + 1.0) \
                + noise * torch.randn(nb, G.CLIP, device=dev, generator=gl)
            feats = conv.ce(G.spectrogram(clips))                           # [nb, 768, 24]
            toks[:, s0 * fpc:(s0 + nb) * fpc] = feats[:, :, :fpc].permute(1, 0, 2).reshape(768, nb * fpc)
        torch.cuda.synchronize()
        t_enc = time.perf_counter() - t0
        dstats = {}
        keep = G.dedup_mask(toks, 0.98, stats=dstats)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        n = toks.shape[1]
        return {"vectors": n, "kept_after_dedup": int(keep.sum()), "encode_s": round(t_enc, 3), "dedup_s": round(t_all - t_enc, 3),
                "vectors_per_s": round(n / t_all, 1), "frames_encoded": n_clips * 24, "corpus_noise": noise,
                "dedup": dstats,
                "peak_device_bytes_above_resident": int(torch.cuda.max_memory_allocated(dev) - mem0), "resident_bytes_before": int(mem0),
                "note": "content encoder in batches of 1000 clips (24 frames each, 8 kept), greedy --dedup 0.98 through the k = 8 self-search of the bank "
                        "(a dense CE-derived bank: the search runs its bf16 / collect tiers)"}
    if "library_build" in legs:
        extra["library_build"] = guarded(lambda: library_build_leg(25_000))
    # ... and at BASELINE config 4's size: 1 000 000 vectors (125 000 clips), the bank generate_voice_library.py exists for
    # Two corpora: breathy clips (noise 0.3: about the share of near-duplicates a speech corpus has) in the default set, and the 200 k
    # leg's clean two-partial clips (74 % of the frames are near-duplicates of an earlier one: three quarters of the self-search fall
    # through to the exact scan, 52 s in round 6 -- `--legs library_build_1m_dense`, not part of "all")
    if "library_build_1m" in legs:
        extra["library_build_1m"] = guarded(lambda: library_build_leg(125_000, noise=0.3))
    if "library_build_1m_dense" in legs:
        extra["library_build_1m_dense"] = guarded(lambda: library_build_leg(125_000))

    # PCIe-inclusive rate (never `value`): the same step with the windows arriving from pinned host memory and the
    # waveforms returned to it, as the CLI edge does (inference.py:88-94,134)
    if "pcie_inclusive" in legs:
        def leg():
            host_in = windows.cpu().pin_memory()
            host_out = torch.empty_like(host_in).pin_memory()

            def s():
                wdev = host_in.to(dev, non_blocking=True)
                host_out.copy_(conv.convert_windows(wdev, k=args.k, window_batch=args.window_batch), non_blocking=True)
            tp, _ = timed_steps(s, 2)
            return {"value": round(frames_per_step / tp, 1), "unit": "frames/s", "ms_per_step": round(tp * 1e3, 2),
                    "bytes_per_step": int(2 * host_in.numel() * 4)}
        extra["pcie_inclusive"] = guarded(leg)

    # The metric says "@24 kHz": 24 kHz utterances in pinned host memory -> device -> resample to 16 kHz -> peak
    # normalise -> windows -> the step -> centre thirds stitched -> resample to 24 kHz -> host (inference.py:86-142
    # without the file I/O).  Never `value`.
    if "e2e_24k" in legs:
        def leg():
            from module import audio_io
            from module.pipeline import make_windows, stitch
            sr = 24000
            sig16 = synth_signals(args.utterances, int(args.seconds * 16000), dev, seed=100)
            host_in = audio_io.resample(sig16, 16000, sr).cpu().pin_memory()          # [U, seconds * 24000]
            host_out = torch.empty_like(host_in).pin_memory()

            def s():
                wf = audio_io.resample(host_in.to(dev, non_blocking=True), sr, 16000)
                wf = wf / wf.abs().amax(dim=1, keepdim=True)
                wins, total = [], 0
                for u in range(wf.shape[0]):
                    w, total = make_windows(wf[u:u + 1], args.chunk)
                    wins.append(w)
                per = wins[0].shape[0]
                o = conv.convert_windows(torch.cat(wins, 0), k=args.k, window_batch=args.window_batch)
                utt = torch.cat([stitch(o[u * per:(u + 1) * per], total, args.chunk) for u in range(wf.shape[0])], 0)
                host_out.copy_(audio_io.resample(utt, 16000, sr)[:, :host_out.shape[1]], non_blocking=True)
            s()
            te, _ = timed_steps(s, 2)
            audio_s = args.utterances * args.seconds
            return {"ms_per_step": round(te * 1e3, 2), "frames_per_s": round(frames_per_step / te, 1),
                    "useful_frames_per_s": round(useful_frames / te, 1), "rtf": round(te / audio_s, 6),
                    "finite": bool(torch.isfinite(host_out).all()),
                    "path": "24 kHz pinned host -> H2D -> resample 24k->16k -> normalise -> windows -> step -> stitch -> resample 16k->24k -> D2H"}
        extra["e2e_24k"] = guarded(leg)

    small_tokens = None
    if "config2" in legs or "streaming" in legs:
        small_tokens = torch.randn(1, 768, 50_000, device=dev, generator=torch.Generator(device=dev).manual_seed(7))

    # BASELINE config 2: one 10 s utterance at a time (batch 1) against a ~50 k-vector single-speaker library: latency
    if "config2" in legs:
        def leg():
            c2 = Converter(conv.ce, conv.pe, conv.dec, dev).set_library(small_tokens)
            wf = synth_signals(1, 160000, dev, seed=5)
            res, outs = {}, {}
            for mode, share in (("per_window_front_end", None), ("overlap_shared", True)):
                import gc

                def calls(n):
                    lat = []
                    for i in range(n):
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        o_ = c2.convert(wf, chunk=args.chunk, k=args.k, window_batch=args.window_batch, share_overlap=share)
                        torch.cuda.synchronize()
                        lat.append((time.perf_counter() - t1) * 1e3)
                    return lat, o_
                first, _ = calls(4)                                  # warm-up: workspaces, side streams, allocator pools
                # round 5 reported max 40.1 ms against p50 4.19 (per-window mode, 10 calls after 2 warm-up calls) without a cause.  The leg
                # now drops four warm-up calls (the first call of a fresh Converter grows its workspaces and creates its side streams:
                # 5 - 12 ms, reported as first_four_calls_ms) and times 50 calls with the cyclic garbage collector held off and 50 with it
                # on, with the collector's own count of passes in the second loop.  Round 6: no call above 4.8 ms in either loop and no
                # collector pass inside them -- the outlier belonged to the warm-up, not to the steady state
                gc.collect()
                gc.disable()
                try:
                    lat, o = calls(50)
                finally:
                    gc.enable()
                n0 = sum(st["collections"] for st in gc.get_stats())
                lat_gc, _ = calls(50)
                n_gc = sum(st["collections"] for st in gc.get_stats()) - n0
                lat = sorted(lat)
                outs[mode] = o
                res[mode] = {"ms_per_utterance_p50": round(lat[len(lat) // 2], 3), "ms_per_utterance_p99": round(lat[int(0.99 * (len(lat) - 1))], 3),
                             "ms_per_utterance_max": round(lat[-1], 3), "calls": len(lat), "first_four_calls_ms": [round(x, 2) for x in first],
                             "with_python_gc_on": {"p50": round(sorted(lat_gc)[len(lat_gc) // 2], 3), "max": round(max(lat_gc), 3), "gc_passes": n_gc},
                             "rtf": round(lat[len(lat) // 2] * 1e-3 / 10.0, 6)}
            res.update(ms_per_utterance_p50=res["per_window_front_end"]["ms_per_utterance_p50"],
                       finite=bool(torch.isfinite(o).all()), same_samples=bool(torch.equal(*outs.values())),
                       workload="one 10 s utterance (6 windows = 2700 frames), 50 000-vector library, batch 1",
                       note="at batch 1 the step is latency-bound: sharing the front end costs more launches than it saves work, so "
                            "the CLI (share_overlap='auto') shares only from ~40 windows at 50 k vectors / ~18 at 1 M")
            return res
        extra["config2"] = guarded(leg)

    # BASELINE config 5: streaming, 10 ms chunks (-c 160 -b 16: 8-frame ring), per-step device pipeline in one hipGraph
    if "streaming" in legs:
        def leg():
            import numpy as np
            from module.realtime import RealtimeConverter
            chunk, bs, steps = 160, 16, args.stream_steps
            rt = RealtimeConverter(conv.ce, conv.pe, conv.dec, small_tokens, dev, chunk=chunk, buffersize=bs).enable_graph()
            pcm = (np.random.default_rng(0).standard_normal(chunk * (steps + bs + 60)) * 3000).astype(np.int16)
            lat = []
            for s_ in range(steps + bs + 50):
                t1 = time.perf_counter()
                o = rt.step(pcm[s_ * chunk:(s_ + 1) * chunk])       # includes H2D of the ring and D2H of the result
                d = time.perf_counter() - t1
                if o is not None and s_ >= bs + 50:
                    lat.append(d * 1e3)
            lat = np.array(lat)
            return {"p50_ms": round(float(np.percentile(lat, 50)), 3), "p99_ms": round(float(np.percentile(lat, 99)), 3),
                    "mean_ms": round(float(lat.mean()), 3), "rtf": round(float(np.percentile(lat, 50)) / (chunk / 16.0), 4),
                    "steps": int(lat.size), "hipgraph": True,
                    "workload": f"int16 chunks of {chunk} samples (10 ms), ring of {bs} chunks = {chunk * bs // FRAME} frames, 50 000-vector library, host in / host out per step"}
        extra["streaming"] = guarded(leg)
    del small_tokens

    sharded = None
    if dist_on and not args.no_shard_library:
        # BASELINE config 4 (never part of `value`): one fixed global batch (same seed on every rank), windows partitioned
        # over the ranks, the library cut into `world` row slabs -- checked bitwise against the replicated path
        from module.sharded import bench_sharded

        def leg():
            wg = synth_windows(args.utterances, args.seconds, args.chunk, dev, seed=100)
            return bench_sharded(conv, tokens, wg, args.k, world, rank, dev, args.window_batch)
        sharded = guarded(leg)

    cpu = None
    if "cpu_baseline" in legs:
        cpu = cpu_baseline(windows, library.rows, args)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        audio_s = world * args.utterances * args.seconds
        from module import ops as _ops
        decoder_mode = 2 if conv.dec._split_for_this_checkpoint() else _ops.decoder_precision(0)     # (the checkpoint's calibrated mode)
        encoder_mode = _ops.encoder_precision(0)
        dtype_detail = ({"fp6": "fp6 (e2m3)", "fp8": "fp8 (e4m3)"}[library.prefilter] + " MFMA candidate scoring + exact f32 rescoring, every frame "
                      "certified at 7 sigma of its measured stage error (statistical: audited against brute force on every frame of this "
                      "batch, tests/test_gpu_knn_audit.py; ALIVE_KNN_STRICT=1 is the deterministic form)" if library.prefilter in ("fp8", "fp6") else
                      "bf16 MFMA scoring + exact f32 rescoring, certified per frame" + (" (deterministic bound)" if library.strict else "")) + "; encoders (fp32-grade): " + ("fp16 split planes (hi + scaled lo, 3 MFMAs per product, 22 significand bits) for the ConvNeXt pointwise convs, 3-plane split bf16 (6 MFMAs) for the DFT / input / output / classifier GEMMs" if encoder_mode == 1 else "3-plane split-bf16 GEMMs (ALIVE_ENCODER_PRECISION=2)") + "; decoder: " + ("plain fp16 (one MFMA per product, fp32 accumulate and residual streams) for the ConvNeXt pointwise convs, the 256- and 64-channel FilterBlocks' k5 convs and four smaller layers (80.3 of its 96.6 MFLOP per frame) -- decoder waveform RMS error 2.9e-5 on the reference's 450-frame fixture against the 1e-3 bar, ALIVE_DECODER_PRECISION=2 restores split bf16 (5e-6) --, 2-plane split-bf16 for its other GEMMs" if decoder_mode == 1 else "2-plane split-bf16 GEMMs (ALIVE_DECODER_PRECISION=2)") + "; f32 MFMA DFT / strided / small-channel convs; f64 phase scan"
        # <= 200 characters, result-affecting arithmetic first (the driver's record keeps 200)
        knn_s = {"fp6": "kNN exact fp32 top-k via fp6 MFMA candidates (certified)", "fp8": "kNN exact fp32 top-k via fp8 MFMA candidates (certified)"}.get(
            library.prefilter, "kNN exact fp32 top-k via bf16 MFMA candidates (" + ("deterministic" if library.strict else "certified") + ")")
        dtype_short = (("decoder fp16 GEMMs / fp32 acc" if decoder_mode == 1 else "decoder split-bf16 GEMMs (fp32-grade)") + "; " +
                       ("encoders 22-bit fp16 split" if encoder_mode == 1 else "encoders bf16x6 (24-bit)") + "; " + knn_s + "; f32 MFMA small convs; f64 phase scan")
        assert len(dtype_short) <= 200, len(dtype_short)
        line = {
            "metric": "VC frames/sec + RTF @24kHz, 1M-vec library; 1/2/4/8 MI355X",
            "value": round(world * frames_per_step * args.steps / dt, 1),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_short, "dtype_detail": dtype_detail,
            "data": "synthetic",
            "config": {"workload": f"{args.utterances} utterances x {args.seconds:g} s per GPU -> {n_win} windows x "
                                   f"{L // FRAME} frames per step, {M}-vector library (BASELINE config 3/4 shape)",
                       "library_vectors": M, "k": args.k, "windows_per_step_per_gpu": n_win,
                       "frames_per_step_per_gpu": frames_per_step, "parallelism": f"dp{world} over windows, library replicated",
                       "window_batch": args.window_batch},
            "useful_frames_per_s": round(world * useful_frames * args.steps / dt, 1),
            "rtf": round((dt / args.steps) / audio_s, 6),
            "roofline": roofline,
            "decoder_precision_mode": decoder_mode, "encoder_precision_mode": encoder_mode,
            "decoder_precision_calibration": conv.dec.calibration,      # module/decoder.py: fp16 forms kept only within 1.5e-4 of split bf16 on a probe
            "fp16_saturations": _ops.f16_saturations(),      # values left saturated at the end of the run (every batch entry point clears and checks: 0)
            "fp16_fallbacks": _ops.Fp16Guard.fallbacks,      # batches the range guard repeated on bf16 planes (module/ops.py::Fp16Guard): 0
            "roofline_nets": roofline_nets,
            "cpu_baseline": cpu,
        }
        if dist_on:
            line["ranks"] = {"backend": dist.get_backend(), "rccl_ranks" if args.backend == "nccl" else "gloo_ranks": dist.get_world_size(),
                             "same_device": bool(args.same_device)}
        line.update(extra)
        if sharded is not None:
            line["sharded_knn"] = sharded
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
