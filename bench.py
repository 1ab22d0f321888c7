"""bench.py -- voice-conversion throughput of the MI355X hot path (contract: see the task prompt).

One "step" = one pass of the whole hot path over one batch of synthetic input that is already
resident in HBM: `--utterances` x `--seconds` of 16 kHz audio cut into the reference's overlapping
windows (inference.py:94-101) -> magnitude STFT -> F0 estimator (+ per-window pitch transform) ->
content encoder -> kNN match against an M-vector voice library -> decoder -> waveform windows.
Default workload = BASELINE.json config 3/4 on one GPU: 64 utterances x 10 s (384 windows,
172 800 content frames per step) against a 1 M-vector library.

N > 1 (launched by torch.distributed.run, one rank per GPU): windows are independent units, so
ranks are pure data-parallel replicas over disjoint utterance sets ("weak" scaling, no data-path
collective); `--shard-library` additionally times the library-sharded kNN (per-shard exact top-k,
RCCL all-gather, merge) that the north star names as the optional mode.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "alive-vc_amd"))

import torch                                              # noqa: E402
import torch.distributed as dist                           # noqa: E402

FRAME = 320
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_FP8_TFLOPS = 5000.0           # dense fp8 MFMA peak (block-scaled 32x32x64 e4m3), same table


def synth_windows(n_utt, seconds, chunk, device, seed):
    """n_utt synthetic utterances (noise + a few partials, peak-normalised like inference.py:92) -> all windows."""
    from module.pipeline import make_windows
    g = torch.Generator(device=device).manual_seed(seed)
    L = int(seconds * 16000)
    t = torch.arange(L, device=device, dtype=torch.float32) / 16000.0
    wins = []
    for u in range(n_utt):
        f = 90.0 + 7.0 * u
        x = 0.1 * torch.randn(L, device=device, generator=g)
        for h in (1, 2, 3):
            x = x + 0.2 / h * torch.sin(2 * torch.pi * f * h * t)
        x = (x / x.abs().max()).view(1, L)
        w, _ = make_windows(x, chunk)
        wins.append(w)
    return torch.cat(wins, 0).contiguous()


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
    ap.add_argument("--window-batch", type=int, default=128)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU-oracle leg (0 = skip)")
    ap.add_argument("--shard-library", action="store_true", help="(default when --gpus > 1) also time the library-sharded kNN + all-gather")
    ap.add_argument("--no-shard-library", action="store_true", help="skip the library-sharded kNN leg of a multi-GPU run")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo + --same-device only exercise the multi-rank code path on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true", help="test only: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device("cuda", 0 if args.same_device else local)
    torch.cuda.set_device(dev)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

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
    del tokens
    conv = Converter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), dev).set_library(library)
    windows = synth_windows(args.utterances, args.seconds, args.chunk, dev, seed=100 + rank)
    n_win, L = windows.shape
    frames_per_step = n_win * (L // FRAME)
    useful_frames = args.utterances * int(args.seconds * 16000) // FRAME

    # ---- per-launch timing of the dominant kernel (the MFMA candidate scoring) with events on its stream ----
    ev_pairs = []
    orig_search = library.search

    def timed_search(src, k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()                               # materialise the handles
        nat.lib().alive_knn_set_timing_events(a.cuda_event, b.cuda_event)
        ev_pairs.append((a, b, src.shape[0] * src.shape[2]))
        try:
            return orig_search(src, k)
        finally:
            nat.lib().alive_knn_set_timing_events(None, None)
    library.search = timed_search

    def step():
        return conv.convert_windows(windows, k=args.k, window_batch=args.window_batch)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    ev_pairs.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    assert torch.isfinite(out).all(), "non-finite waveform"
    researched = library.fallback_frames() if library.prefilter == "fp8" else None     # of the last timed step's search

    # ---- roofline of the scoring kernel ----
    flops, ms = 0.0, 0.0
    for a, b, tt_frames in ev_pairs:
        ms += a.elapsed_time(b)
        flops += 2.0 * 768 * M * tt_frames
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    traffic = None
    fp8 = library.prefilter == "fp8"
    peak = PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS
    pmc = os.path.join(ROOT, "profiles", "knn_score8_pmc.json" if fp8 else "knn_score_pmc.json")
    if os.path.exists(pmc):
        traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
    roofline = {"kernel": "knn_score8_kernel" if fp8 else "knn_score_kernel", "bound": "mfma", "achieved": round(achieved, 1),
                "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "mfma_dtype": "fp8 e4m3, block-scaled 32x32x64" if fp8 else "bf16 32x32x16",
                "launches": len(ev_pairs), "avg_launch_ms": round(ms / max(1, len(ev_pairs)), 3),
                "kernel_share_of_step": round(ms * 1e-3 / dt, 3)}

    # Data dependence of the scoring kernel: the candidate-list path is taken more often when the frames of a wave
    # are uncorrelated.  The batch above comes from one synthetic signal family (correlated frames); time the same
    # kernel once on i.i.d. Gaussian frames so both ends are on record (real speech lies in between).
    ev_pairs.clear()
    rnd = torch.randn(n_win, 768, L // FRAME, device=dev, generator=torch.Generator(device=dev).manual_seed(77 + rank))
    library.search(rnd, args.k)
    torch.cuda.synchronize()
    a, b, tt_frames = ev_pairs[-1]
    roofline["achieved_uncorrelated_frames"] = round(2.0 * 768 * M * tt_frames / (a.elapsed_time(b) * 1e-3) / 1e12, 1)
    del rnd
    if fp8:            # frames of the last timed batch whose fp8 candidate set was not certified and went through the bf16 stage
        roofline["frames_researched_on_bf16"] = {"timed_batch": researched, "uncorrelated_batch": library.fallback_frames(),
                                                 "of": frames_per_step}

    # Optional mode of the build, reported beside the headline and never as `value`: inference.py keeps the centre third of
    # every window, so only the frames that can reach it through the decoder need the kNN match (Converter(keep_frames=...),
    # `--trim-context`).  The kept samples are bitwise those of the full computation (checked here on the whole batch: kept_samples_bitwise_equal).
    trim = None
    if rank == 0 and world == 1:
        try:                                                  # a failure here must not cost the headline line
            library.search = orig_search
            cf = (L // FRAME) // 3
            ref_out = out[:, cf * FRAME:2 * cf * FRAME].clone()
            conv.convert_windows(windows, k=args.k, window_batch=args.window_batch, keep_frames=(cf, 2 * cf))    # warm-up (scratch sizes)
            torch.cuda.synchronize()
            tt0 = time.perf_counter()
            for _ in range(2):
                out_t = conv.convert_windows(windows, k=args.k, window_batch=args.window_batch, keep_frames=(cf, 2 * cf))
            torch.cuda.synchronize()
            tt0 = (time.perf_counter() - tt0) / 2
            same = bool(torch.equal(out_t[:, cf * FRAME:2 * cf * FRAME], ref_out))
            trim = {"ms_per_step": round(tt0 * 1e3, 2), "windows_per_s": round(n_win / tt0, 1),
                    "useful_frames_per_s": round(useful_frames / tt0, 1), "kept_samples_bitwise_equal": same,
                    "note": "kNN match and decoder on frames [cf-32, 2cf+16) of each window (oscillator phase over the whole window), content encoder on that range +-16; spectrogram and f0 estimator on the whole window"}
            del out_t, ref_out
        except Exception as e:
            trim = {"error": f"{type(e).__name__}: {e}"[:300]}

    # PCIe-inclusive rate (never `value`): the same step with the windows arriving from pinned host memory and the
    # waveforms returned to it, as the CLI edge does (inference.py:88-94,134)
    pcie = None
    if rank == 0 and world == 1:
        try:
            library.search = orig_search
            host_in = windows.cpu().pin_memory()
            host_out = torch.empty_like(host_in).pin_memory()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(2):
                wdev = host_in.to(dev, non_blocking=True)
                host_out.copy_(conv.convert_windows(wdev, k=args.k, window_batch=args.window_batch), non_blocking=True)
            torch.cuda.synchronize()
            tp = (time.perf_counter() - tp) / 2
            pcie = {"value": round(frames_per_step / tp, 1), "unit": "frames/s", "ms_per_step": round(tp * 1e3, 2),
                    "bytes_per_step": int(2 * host_in.numel() * 4)}
            del host_in, host_out, wdev
        except Exception as e:
            pcie = {"error": f"{type(e).__name__}: {e}"[:300]}

    sharded = None
    if world > 1 and not args.no_shard_library:
        # BASELINE config 4 (never part of `value`): the library cut into `world` row slabs, every rank scores one window
        # batch against its slab, exact per-shard top-k merged after one all-gather over RCCL
        from module.sharded import bench_sharded_knn
        library.search = orig_search
        try:
            sharded = bench_sharded_knn(conv, windows[: args.window_batch], M, args.k, world, rank, dev)
        except Exception as e:                                # keep the headline line even if this leg fails
            sharded = {"error": f"{type(e).__name__}: {e}"[:300]}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        library.search = orig_search
        cpu = cpu_baseline(windows, library.rows, args)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        audio_s = world * args.utterances * args.seconds
        line = {
            "metric": "VC frames/sec + RTF @24kHz, 1M-vec library; 1/2/4/8 MI355X",
            "value": round(world * frames_per_step * args.steps / dt, 1),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("fp8 (e4m3) MFMA candidate scoring, certified, + exact f32 rescoring" if library.prefilter == "fp8" else
                      "bf16 MFMA scoring + exact f32 rescoring") + "; 3-plane split-bf16 (fp32-grade) encoder GEMMs; 2-plane split-bf16 decoder GEMMs; f32 MFMA DFT / strided / small-channel convs; f64 phase scan",
            "data": "synthetic",
            "config": {"workload": f"{args.utterances} utterances x {args.seconds:g} s per GPU -> {n_win} windows x "
                                   f"{L // FRAME} frames per step, {M}-vector library (BASELINE config 3/4 shape)",
                       "library_vectors": M, "k": args.k, "windows_per_step_per_gpu": n_win,
                       "frames_per_step_per_gpu": frames_per_step, "parallelism": f"dp{world} over windows, library replicated",
                       "window_batch": args.window_batch},
            "useful_frames_per_s": round(world * useful_frames * args.steps / dt, 1),
            "rtf": round((dt / args.steps) / audio_s, 6),
            "roofline": roofline,
            "cpu_baseline": cpu,
            "pcie_inclusive": pcie,
            "context_trim": trim,
        }
        if sharded is not None:
            line["sharded_knn"] = sharded
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
