"""one batch (128 windows x 450 frames) through spectrogram + f0 estimator + content encoder + decoder, no kNN:
the workload of the PMC passes over the network kernels (tools/pmc_nets.sh)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import synth_windows
from module.content_encoder import ContentEncoder
from module.decoder import Decoder
from module.f0_estimator import F0Estimator
from module.spectrogram import spectrogram
dev = "cuda"
ce, pe, dec = ContentEncoder(seed=2).to(dev), F0Estimator(seed=2).to(dev), Decoder(seed=2).to(dev)
w = synth_windows(22, 10.0, 48000, dev, 100)[:128].contiguous()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for _ in range(reps):
    spec = spectrogram(w)
    f0 = pe.estimate(spec)
    feat = ce(spec)
    wave, _ = dec(feat, f0=f0 * 0.05 + 100.0)
torch.cuda.synchronize()
print("ok", tuple(wave.shape))
