"""Offline conversion CLI -- same flags, defaults and file naming as the reference's inference.py
(/root/reference/inference.py:20-43,141), running the whole hot path on the MI355X.

Differences, all at the unpinned edges (DESIGN.md "out of scope / unpinned"): WAV I/O and resampling
are this package's own (torchaudio is not available), output files are float32 WAV, and the two mel
PNG plots the reference writes are skipped unless matplotlib is importable AND --plots is given.
-d/--device must name a HIP device ("cuda", the reference's own spelling for ROCm); there is no
CPU execution path here.
"""
import argparse
import glob
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from module import audio_io                                     # noqa: E402
from module.content_encoder import ContentEncoder                # noqa: E402
from module.decoder import Decoder                               # noqa: E402
from module.f0_estimator import F0Estimator                      # noqa: E402
from module.pipeline import Converter                            # noqa: E402
from module.spectrogram import spectrogram                       # noqa: E402
from module.voice_library import VoiceLibrary                    # noqa: E402


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('-i', '--inputs', default="./inputs/")
    parser.add_argument('-o', '--outputs', default="./outputs/")
    parser.add_argument('-dep', '--decoder-path', default="decoder.pt")
    parser.add_argument('-disp', '--discriminator-path', default="discriminator.pt")
    parser.add_argument('-cep', '--content-encoder-path', default="content_encoder.pt")
    parser.add_argument('-f0ep', '--f0-estimator-path', default="f0_estimator.pt")
    parser.add_argument('-f0', '--f0-rate', default=1.0, type=float)
    parser.add_argument('-p', '--pitch', default=0, type=float)
    parser.add_argument('-int', '--intonation', default=1.0, type=float)
    parser.add_argument('-t', '--target', default='NONE')
    parser.add_argument('-d', '--device', default='cuda')
    parser.add_argument('-g', '--gain', default=1.0, type=float)
    parser.add_argument('-a', '--alpha', default=0.0, type=float)
    parser.add_argument('-k', default=4, type=int)
    parser.add_argument('--knn-strict', action='store_true',
                        help="kNN match with the deterministic certificate (bf16 candidates under a Cauchy-Schwarz error bound; "
                             "about twice the search time; this build only; same as ALIVE_KNN_STRICT=1)")
    parser.add_argument('-c', '--chunk', default=48000, type=int)
    parser.add_argument('-lib', '--voice-library-path', default="NONE")
    parser.add_argument('-noise', '--noise-amp', default=1.0, type=float)        # parsed and unused, as in the reference
    parser.add_argument('-harmonics', '--harmonics-amp', default=1.0, type=float)
    parser.add_argument('-pf', '--post-filter-alpha', default=0.0, type=float)
    parser.add_argument('-wpe', '--world-pitch-estimation', default=False)
    parser.add_argument('-norm', '--normalize', default=False, type=bool)
    parser.add_argument('--window-batch', default=64, type=int, help="windows per device batch (this build only)")
    parser.add_argument('--trim-context', action='store_true',
                        help="(default since round 5, kept for compatibility) match and decode only the frames that can reach the kept "
                             "centre third of each window: same samples, ~44 %% of the kNN and decoder work")
    parser.add_argument('--no-trim-context', action='store_true',
                        help="run the kNN match and the decoder over all three chunks of every window as the reference does (same "
                             "samples in the kept centre third either way; this build only)")
    parser.add_argument('--no-share-overlap', action='store_true',
                        help="always run spectrogram / f0 estimator / content encoder / kNN match per window as the reference does; by "
                             "default long utterances run them once per utterance with the windows assembled from it (same samples "
                             "either way; this build only)")
    parser.add_argument('--pcm16', action='store_true', help="write 16-bit PCM instead of float32 WAV (this build only)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.knn_strict:
        os.environ["ALIVE_KNN_STRICT"] = "1"              # read by module/common.py when the library is packed
    device = torch.device(args.device)
    if device.type != "cuda":
        raise SystemExit("this build runs on the MI355X only: pass -d cuda (the reference's spelling for ROCm devices)")
    if args.world_pitch_estimation:
        raise SystemExit("-wpe needs pyworld (WORLD), which is outside this build's scope")

    PE, CE, Dec = F0Estimator().to(device), ContentEncoder().to(device), Decoder().to(device)
    PE.load_state_dict(torch.load(args.f0_estimator_path, map_location=device))
    CE.load_state_dict(torch.load(args.content_encoder_path, map_location=device))
    Dec.load_state_dict(torch.load(args.decoder_path, map_location=device))
    os.makedirs(args.outputs, exist_ok=True)

    tgt = torch.zeros(1, 768, 0, device=device)
    if args.target != "NONE":
        print("target: encoding the reference wav into library frames")
        wf, sr = audio_io.load(args.target)
        wf = audio_io.resample(wf.to(device), sr, 16000)
        wf = wf / wf.abs().max()
        wf = wf[:1]
        tgt = CE(spectrogram(wf))
    if args.voice_library_path != "NONE":
        print(f"target: voice library file {args.voice_library_path}")
        VL = VoiceLibrary().to(device)
        VL.load_state_dict(torch.load(args.voice_library_path, map_location=device))
        tgt = torch.cat([tgt, VL.tokens], dim=2)
    print(f"library holds {tgt.shape[2]} vectors")
    conv = Converter(CE, PE, Dec, device).set_library(tgt)

    paths = glob.glob(os.path.join(args.inputs, "*"))        # the reference's enumeration (inference.py:86): its order decides the {i}_ prefix of the outputs
    for i, path in enumerate(paths):
        wf, sr = audio_io.load(path)
        wf = audio_io.resample(wf.to(device), sr, 16000)
        wf = wf / wf.abs().max()
        wf = wf.mean(dim=0, keepdim=True)
        print(f"-> {path}")
        out = conv.convert(wf, chunk=args.chunk, k=args.k, alpha=args.alpha, pitch_shift=args.pitch,
                           intonation=args.intonation, f0_rate=args.f0_rate, window_batch=args.window_batch,
                           trim_context=not args.no_trim_context,
                           share_overlap=None if args.no_share_overlap else "auto")
        out = audio_io.resample(out, 16000, sr, post_gain_db=args.gain).cpu()      # resample, then gain (:136-137)
        if args.normalize:
            out = out / out.abs().max()
        file_name = f"{i}_{os.path.splitext(os.path.basename(path))[0]}"
        audio_io.save(os.path.join(args.outputs, f"{file_name}.wav"), out, sr, "pcm16" if args.pcm16 else "float32")


if __name__ == "__main__":
    main()
