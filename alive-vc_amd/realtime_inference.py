"""Streaming CLI -- flags and defaults of the reference's realtime_inference.py:20-52.

With PyAudio installed it opens the same int16 input/output (and optional loopback) streams as the
reference; without it (this image) use --input-wav / --output-wav to stream a file through the
identical per-chunk loop.  -d must be "cuda" (the reference's spelling for ROCm).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from module import audio_io                                     # noqa: E402
from module.content_encoder import ContentEncoder                # noqa: E402
from module.decoder import Decoder                               # noqa: E402
from module.f0_estimator import F0Estimator                      # noqa: E402
from module.realtime import RealtimeConverter                    # noqa: E402
from module.spectrogram import spectrogram                       # noqa: E402
from module.voice_library import VoiceLibrary                    # noqa: E402


def build_parser():
    parser = argparse.ArgumentParser(description="Convert voice")
    parser.add_argument('-d', '--device', default='cuda', choices=['cpu', 'cuda', 'mps'],
                        help="Compute device setting. Set this option to cuda if you need to use ROCm.")
    parser.add_argument('-i', '--input', default=0, type=int)
    parser.add_argument('-o', '--output', default=0, type=int)
    parser.add_argument('-l', '--loopback', default=-1, type=int)
    parser.add_argument('-g', '--gain', default=0.0, type=float)
    parser.add_argument('-ig', '--input-gain', default=0.0, type=float)
    parser.add_argument('-dep', '--decoder-path', default="decoder.pt")
    parser.add_argument('-cep', '--content-encoder-path', default="content_encoder.pt")
    parser.add_argument('-f0ep', '--f0-estimator-path', default="f0_estimator.pt")
    parser.add_argument('-b', '--buffersize', default=8, type=int)
    parser.add_argument('-c', '--chunk', default=960, type=int)
    parser.add_argument('-ic', '--inputchannels', default=1, type=int)
    parser.add_argument('-oc', '--outputchannels', default=1, type=int)
    parser.add_argument('-lc', '--loopbackchannels', default=1, type=int)
    parser.add_argument('-f0', '--f0-rate', default=1, type=float)
    parser.add_argument('-p', '--pitch', default=0, type=float)
    parser.add_argument('-t', '--target', default='NONE')
    parser.add_argument('-k', default=4, type=int)
    parser.add_argument('--knn-strict', action='store_true',
                        help="kNN match with the deterministic certificate (bf16 candidates under a Cauchy-Schwarz error bound; "
                             "about twice the search time; this build only; same as ALIVE_KNN_STRICT=1)")
    parser.add_argument('-a', '--alpha', default=0.0, type=float)
    parser.add_argument('-fp16', default=False, type=bool)
    parser.add_argument('-lib', '--voice-library-path', default="NONE")
    parser.add_argument('-wpe', '--world-pitch-estimation', default=False, type=bool)
    parser.add_argument('-isr', '--input-sr', default=16000, type=int)
    parser.add_argument('-osr', '--output-sr', default=16000, type=int)
    parser.add_argument('-lsr', '--loopback-sr', default=16000, type=int)
    parser.add_argument('--input-wav', default=None, help="stream this file instead of an audio device (this build only)")
    parser.add_argument('--output-wav', default=None, help="write the converted stream here (this build only)")
    parser.add_argument('--no-graph', action='store_true',
                        help="launch the per-chunk device pipeline kernel by kernel instead of replaying one captured hipGraph (this build only)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.knn_strict:
        os.environ["ALIVE_KNN_STRICT"] = "1"              # read by module/common.py when the library is packed
    if args.device != 'cuda' or not torch.cuda.is_available():
        raise SystemExit("Error: this build needs a ROCm device: pass -d cuda on an MI355X host.")
    if args.fp16 or args.world_pitch_estimation:
        raise SystemExit("-fp16 (documented deprecated upstream) and -wpe (pyworld) are outside this build's scope")
    device = torch.device('cuda')
    PE, CE, Dec = F0Estimator().to(device), ContentEncoder().to(device), Decoder().to(device)
    PE.load_state_dict(torch.load(args.f0_estimator_path, map_location=device))
    CE.load_state_dict(torch.load(args.content_encoder_path, map_location=device))
    Dec.load_state_dict(torch.load(args.decoder_path, map_location=device))

    tgt = torch.zeros(1, 768, 0, device=device)
    if args.target != "NONE":
        print("target: encoding the reference wav into library frames")
        wf, sr = audio_io.load(args.target)
        wf = audio_io.resample(wf.to(device), sr, 16000)
        wf = wf / wf.abs().max()
        tgt = CE(spectrogram(wf[:1]))[:, :, ::4]                # the realtime script subsamples the target (:88)
    if args.voice_library_path != "NONE":
        print(f"target: voice library file {args.voice_library_path}")
        VL = VoiceLibrary().to(device)
        VL.load_state_dict(torch.load(args.voice_library_path, map_location=device))
        tgt = torch.cat([tgt, VL.tokens], dim=2)
    print(f"library holds {tgt.shape[2]} vectors")

    rt = RealtimeConverter(CE, PE, Dec, tgt.contiguous(), device, chunk=args.chunk, buffersize=args.buffersize,
                           input_sr=args.input_sr, output_sr=args.output_sr, f0_rate=args.f0_rate, pitch=args.pitch,
                           k=args.k, alpha=args.alpha, gain=args.gain, input_gain=args.input_gain)
    if not args.no_graph:
        rt.enable_graph()        # the whole per-chunk device pipeline (~150 launches) captured once, replayed per chunk: same samples
    print("streaming: conversion running (Ctrl-C stops)")
    if args.input_wav is not None:
        wf, sr = audio_io.load(args.input_wav)
        wf = audio_io.resample(wf.mean(dim=0, keepdim=True).to(device), sr, args.input_sr)[0].cpu()
        pcm = (wf.numpy() * 32767).astype(np.int16)
        outs = []
        for s in range(0, len(pcm) - args.chunk + 1, args.chunk):
            o = rt.step(pcm[s:s + args.chunk])
            if o is not None:
                outs.append(o)
        if args.output_wav and outs:
            audio_io.save(args.output_wav, torch.from_numpy(np.concatenate(outs).astype(np.float32) / 32768)[None],
                          args.output_sr, "pcm16")
        return
    try:
        import pyaudio
    except ImportError:
        raise SystemExit("PyAudio is not installed: use --input-wav/--output-wav to stream a file")
    audio = pyaudio.PyAudio()
    sin = audio.open(format=pyaudio.paInt16, rate=args.input_sr, channels=args.inputchannels, input_device_index=args.input, input=True)
    sout = audio.open(format=pyaudio.paInt16, rate=args.output_sr, channels=args.outputchannels, output_device_index=args.output, output=True)
    sloop = audio.open(format=pyaudio.paInt16, rate=args.loopback_sr, channels=args.loopbackchannels,
                       output_device_index=args.loopback, output=True) if args.loopback != -1 else None
    while True:
        data = np.frombuffer(sin.read(args.chunk), dtype=np.int16)
        out = rt.step(data)
        if out is None:
            continue
        sout.write(out.tobytes())
        if sloop is not None:
            sloop.write(out.tobytes())


if __name__ == "__main__":
    main()
