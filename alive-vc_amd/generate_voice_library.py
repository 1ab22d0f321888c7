"""Voice-library builder -- same positional/flags and on-disk format as the reference's
generate_voice_library.py:13-43 (torch.save({'tokens': float32[1, 768, M]})).

The reference fills 512 fixed slots with one random content frame from each of up to 513 random
7680-sample clips (unseeded).  This build batches the content encoder over all clips on the
MI355X, takes --frames-per-clip frames per clip, is seedable, and writes any M (--num-tokens,
default 512 so the file stays loadable by the reference's VoiceLibrary()).  --dedup COS drops frames whose cosine
to an earlier kept frame exceeds COS (silence and sustained vowels fill real corpora with near-duplicates that only
dilute a k = 4 match); the scan is the library's own kNN kernel run against itself.
"""
import argparse
import glob
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from module import audio_io, ops                                # noqa: E402
from module.content_encoder import ContentEncoder                # noqa: E402
from module.spectrogram import spectrogram                       # noqa: E402
from module.voice_library import VoiceLibrary                    # noqa: E402

CLIP = 7680          # WaveFileDirectory(length=7680): 24 content frames per clip


def collect_clips(root, max_clips, rng):
    """all non-overlapping 7680-sample clips of every wav under `root`, 16 kHz mono, peak-normalised per file
    (module/dataset.py:25-40 of the reference)."""
    clips = []
    for path in sorted(glob.glob(os.path.join(root, "**", "*.wav"), recursive=True)):
        wf, sr = audio_io.load(path)
        wf = audio_io.resample(wf.to("cuda"), sr, 16000).cpu()          # device resampler (csrc/audio.hip)
        wf = wf / wf.abs().max().clamp(min=1e-8)
        wf = wf.mean(dim=0)
        n = wf.shape[0] // CLIP
        if n:
            clips.append(wf[: n * CLIP].view(n, CLIP))
    if not clips:
        raise SystemExit(f"no usable .wav files under {root}")
    clips = torch.cat(clips, 0)
    order = list(range(clips.shape[0]))
    rng.shuffle(order)
    return clips[order[:max_clips]]


def dedup_mask(tokens_DxM, threshold, k=8, chunk=65536, stats=None):
    """keep[m] = False when some KEPT frame m' < m has cos(m, m') > threshold (greedy in index order).  Neighbours come
    from the HIP kNN search of the library against itself (k nearest, the frame itself included), in chunks of frames;
    the greedy recursion is resolved on the device by passes (alive_dedup_pass): a frame is decided once all of its
    earlier near neighbours are -- as many passes as the longest chain of near-duplicates, no per-row host loop (the
    undecided count is read back once per 8 passes).  stats: a dict that receives the time of the self-search, the number
    of passes and what the search's tiers did with the last chunk."""
    import time
    from module import _native as nat
    from module.common import PackedLibrary
    m = tokens_DxM.shape[1]
    k = min(k, m)
    dev = tokens_DxM.device
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    lib = PackedLibrary(tokens_DxM.contiguous())
    val = torch.empty(m, k, dtype=torch.float32, device=dev)
    idx = torch.empty(m, k, dtype=torch.int32, device=dev)
    exact = 0
    for s in range(0, m, chunk):
        v, i = lib.search(tokens_DxM[:, s:s + chunk].unsqueeze(0).contiguous(), k)
        val[s:s + chunk], idx[s:s + chunk] = v, i
        if stats is not None:
            exact += int(lib.search_stats().get("frames_searched_exactly", 0))
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    state = torch.zeros(m, dtype=torch.int32, device=dev)
    undecided = torch.zeros(1, dtype=torch.int32, device=dev)
    passes = 0
    while passes <= m:
        for _ in range(8):
            undecided.zero_()
            nat.check(nat.lib().alive_dedup_pass(nat.ptr(val), nat.ptr(idx), m, k, float(threshold), nat.ptr(state), nat.ptr(undecided),
                                                 nat.stream()), "alive_dedup_pass")
        passes += 8
        if int(undecided.item()) == 0:
            break
    if stats is not None:
        torch.cuda.synchronize(dev)
        stats.update(self_search_s=round(t1 - t0, 3), passes_s=round(time.perf_counter() - t1, 3), passes=passes,
                     frames_searched_exactly=exact, prefilter=lib.prefilter, last_chunk_tiers=lib.search_stats())
    return (state == 1).cpu()


def main(argv=None):
    parser = argparse.ArgumentParser(description="Generate voice library from wave files")
    parser.add_argument("dataset")
    parser.add_argument("-lib", "--voice-library-path", default="voice_library.pt")
    parser.add_argument('-cep', '--content-encoder-path', default="content_encoder.pt")
    parser.add_argument('--num-tokens', default=512, type=int)
    parser.add_argument('--frames-per-clip', default=1, type=int)
    parser.add_argument('--seed', default=None, type=int)
    parser.add_argument('--dedup', default=None, type=float, metavar="COS",
                        help="drop a frame when an earlier frame of the library has cosine similarity above COS")
    parser.add_argument('-d', '--device', default='cuda')
    args = parser.parse_args(argv)
    rng = random.Random(args.seed)
    device = torch.device(args.device)
    CE = ContentEncoder().to(device)
    CE.load_state_dict(torch.load(args.content_encoder_path, map_location=device))
    VL = VoiceLibrary(args.num_tokens)
    if args.seed is not None:
        VL.tokens = torch.randn(1, 768, args.num_tokens, generator=torch.Generator().manual_seed(args.seed))
    print("encoding the corpus into library frames")
    need = (args.num_tokens + args.frames_per_clip - 1) // args.frames_per_clip
    clips = collect_clips(args.dataset, need, rng)
    filled = 0
    for s in range(0, clips.shape[0], 256):
        # under the fp16 range guard: a batch whose activations leave fp16's range is encoded again on bf16 planes BEFORE any of
        # its frames enters the bank (module/ops.py::Fp16Guard)
        feats = ops.Fp16Guard().run(lambda: CE(spectrogram(clips[s:s + 256].to(device)))).cpu()          # [B, 768, 24]
        for b in range(feats.shape[0]):
            # the reference draws frame randint(0, 7) of each clip (generate_voice_library.py:37)
            frames = rng.sample(range(0, 8), min(args.frames_per_clip, 8))
            for fr in frames:
                if filled < args.num_tokens:
                    VL.tokens[0, :, filled] = feats[b, :, fr]
                    filled += 1
    if args.dedup is not None and filled > 1:
        keep = dedup_mask(VL.tokens[0, :, :filled].to(device), args.dedup)
        kept = VL.tokens[0, :, :filled][:, keep.cpu()]
        print(f"dedup: {filled - kept.shape[1]} of {filled} frames have an earlier neighbour above cos {args.dedup}")
        VL.tokens = torch.cat([kept, VL.tokens[0, :, filled:]], 1).unsqueeze(0).contiguous()
        filled = kept.shape[1]
    print(f"saving ({filled} of {VL.tokens.shape[2]} slots from data)")
    torch.save(VL.state_dict(), args.voice_library_path)
    print("done")


if __name__ == "__main__":
    main()
