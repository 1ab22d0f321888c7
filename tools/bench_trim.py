"""stage timing of one step (384 windows, 1 M library) with and without context trimming"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import synth_windows
from module.common import PackedLibrary
from module.content_encoder import ContentEncoder
from module.decoder import Decoder
from module.f0_estimator import F0Estimator
from module.pipeline import Converter, TRIM_LEFT, TRIM_RIGHT
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
lib = PackedLibrary(torch.randn(768, M, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
conv = Converter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), dev).set_library(lib)
w = synth_windows(64, 10.0, 48000, dev, 100)
n, L = w.shape; lf = L // 320; wb = 128
def t(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
for keep in (None, (150, 300)):
    rng = None if keep is None else (keep[0] - TRIM_LEFT, keep[1] + TRIM_RIGHT)
    def feats():
        f = torch.empty(n, 768, lf, device=dev); f0 = torch.empty(n, 1, lf, device=dev)
        for i in range(0, n, wb): f[i:i + wb], f0[i:i + wb] = conv.features(w[i:i + wb], frames=rng)
        return f, f0
    tf, (f, f0) = t(feats)
    sub = f if rng is None else f[:, :, rng[0]:rng[1]].contiguous()
    tm, m = t(lambda: conv.match(sub))
    if rng is None:
        td, _ = t(lambda: [conv.dec(m[i:i + wb], f0[i:i + wb]) for i in range(0, n, wb)])
    else:
        td, _ = t(lambda: [conv.dec.forward_range(m[i:i + wb], f0[i:i + wb], rng[0]) for i in range(0, n, wb)])
    tw, _ = t(lambda: conv.convert_windows(w, window_batch=wb, keep_frames=keep))
    print(f"keep {keep}: features {tf:.1f} ms  match {tm:.1f} ms  decode {td:.1f} ms  sum {tf + tm + td:.1f}  convert_windows {tw:.1f} ms")
