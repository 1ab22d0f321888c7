"""fp8 candidate stage against the bf16 one: same exact top-k?  python tools/knn_fp8_check.py [N] [T] [M] [kind]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
from module.common import PackedLibrary
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 450
M = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
kind = sys.argv[4] if len(sys.argv) > 4 else "randn"
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
tok = torch.randn(768, M, device=dev, generator=g)
src = torch.randn(N, 768, T, device=dev, generator=g)
if kind == "biased":
    src = src * 0.2 + torch.randn(1, 768, 1, device=dev, generator=g)
    tok = tok * 0.5 + torch.randn(768, 1, device=dev, generator=g)           # dense top: cosines cluster high
# fp8 conversion check against torch's e4m3fn cast
x = torch.randn(4096, device=dev) * 0.05
lib16, lib8 = PackedLibrary(tok, prefilter="bf16"), PackedLibrary(tok, prefilter="fp8")
ref8 = (lib16.lib_bf16[:64].float() * 256).to(torch.float8_e4m3fn).view(torch.uint8)
got8 = lib8.lib_f8[:64 * 768].view(64, 768)
print("fp8 pack equals torch e4m3fn cast:", bool(torch.equal(ref8, got8)), "mismatches", int((ref8 != got8).sum()))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); b.record()
res = {}
for name, lib in (("bf16", lib16), ("fp8", lib8)):
    lib.search(src, 4)
    v, i = lib.search(src, 4, events=(a, b))
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    res[name] = (v, i)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); lib.search(src, 4); t1.record(); torch.cuda.synchronize()
    print(f"{name}: scoring kernel {ms:.3f} ms  {2*768*M*N*T/ms/1e9:.1f} TFLOP/s; whole search {t0.elapsed_time(t1):.2f} ms; re-searched frames {lib.fallback_frames()}")
(v0, i0), (v1, i1) = res["bf16"], res["fp8"]
same = (i0 == i1).all(dim=1)
print(f"frames {N*T}: identical ordered top-4 in {int(same.sum())}  differing {int((~same).sum())}")
if (~same).any():
    bad = (~same).nonzero()[:5, 0]
    for f in bad.tolist():
        print(f, v0[f].tolist(), i0[f].tolist(), v1[f].tolist(), i1[f].tolist())
