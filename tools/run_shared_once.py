"""the bench batch through the overlap-sharing path only (for rocprofv3): python tools/run_shared_once.py [passes]"""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "alive-vc_amd")); sys.path.insert(0, ROOT)
import bench
from module.common import PackedLibrary
from module.content_encoder import ContentEncoder
from module.decoder import Decoder
from module.f0_estimator import F0Estimator
from module.pipeline import Converter
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda")
lib = PackedLibrary(torch.randn(768, 1_000_000, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
conv = Converter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), dev).set_library(lib)
windows = bench.synth_windows(64, 10.0, 48000, dev, seed=100)
for _ in range(passes):
    out = conv.convert_windows(windows, k=4, window_batch=128, share_overlap=6)
torch.cuda.synchronize()
print("frames through the front end per pass:", conv.last_front_end_frames, "finite:", bool(torch.isfinite(out).all()))
