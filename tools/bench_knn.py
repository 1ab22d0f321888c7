"""stand-alone kNN search timing (the scoring kernel dominates): python tools/bench_knn.py [N] [T] [M] [reps] [scale]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
from module.common import PackedLibrary
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 450
M = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
kind = sys.argv[5] if len(sys.argv) > 5 else "randn"
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
lib = PackedLibrary(torch.randn(768, M, device=dev, generator=g))
src = torch.randn(N, 768, T, device=dev, generator=g)
if kind == "biased":      # frames that share a large common component, like content-encoder outputs of similar audio
    src = src * 0.2 + torch.randn(1, 768, 1, device=dev, generator=g)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); b.record()
lib.search(src, 4)
tot = 0.0
for _ in range(reps):
    lib.search(src, 4, events=(a, b))
    torch.cuda.synchronize()
    tot += a.elapsed_time(b)
ms = tot / reps
print(f"N {N} T {T} M {M} {kind}: score kernel {ms:.3f} ms  {2*768*M*N*T/ms/1e9:.1f} TFLOP/s")
