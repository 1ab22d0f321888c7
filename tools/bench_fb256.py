"""time of the fused 256-channel FilterBlock alone (csrc/filter_big.hip): python tools/bench_fb256.py [N] [L]
(a library built with `make EXTRA=-DALIVE_FB256_PROF` also prints one block's phase clocks per call)"""
import ctypes, hashlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4500
C = int(sys.argv[3]) if len(sys.argv) > 3 else 256          # 64: alive_filter_block64s_fp16 (L = 36000 for the decoder's shape)
Lf = L // (10 if C == 256 else 80)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(5)
x = 0.3 * torch.randn(N, C, L, device=dev, generator=g); skip = 0.3 * torch.randn(N, C, L, device=dev, generator=g); out = torch.empty_like(x)
film = 0.05 * torch.randn(N, 4128, Lf, device=dev, generator=g)
ws = [(torch.randn(C * 5 * C, device=dev, generator=g) * (0.32 / C ** 0.5)).to(torch.float16) for _ in range(6)]
bs = [torch.randn(C, device=dev, generator=g) * 0.1 for _ in range(6)]
W = (ctypes.c_void_p * 6)(*[w.data_ptr() for w in ws]); B = (ctypes.c_void_p * 6)(*[b.data_ptr() for b in bs])
L_ = nat.lib()
st = torch.cuda.current_stream().cuda_stream
entry = L_.alive_filter_block256_fp16 if C == 256 else L_.alive_filter_block64s_fp16
wsb = (L_.alive_filter_block256_workspace_bytes if C == 256 else L_.alive_filter_block64s_workspace_bytes)(N, L); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
def run(): nat.check(entry(x.data_ptr(), N, L, W, B, film.data_ptr(), 4128, Lf, 0, 0, 0, Lf, skip.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, st))
for _ in range(2): run()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): run()
e.record(); torch.cuda.synchronize()
ms = a.elapsed_time(e) / 5
print(f"{ms:.3f} ms per {N} x {C} x {L}; {6 * 2 * C * 5 * C * N * L / ms / 1e9:.0f} TFLOP/s; finite {bool(torch.isfinite(out).all())} sat {L_.alive_f16_saturations(1)} digest {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}")
