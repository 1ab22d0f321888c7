"""phase stamps of the 8-wave fused 64-channel FilterBlock (diagnostic build: tools/ab_build.sh tools/_ab/fb64_stamps.so filter_mid.hip -DALIVE_STAMPS;
ALIVE_VC_LIB=tools/_ab/fb64_stamps.so python tools/stamp_fb64.py)"""
import sys, os, torch, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
dev = "cuda"; N, L, Lf = 128, 36000, 450
x = torch.randn(N, 64, L, device=dev); skip = torch.randn(N, 64, L, device=dev); out = torch.empty_like(x)
film = torch.randn(N, 4128, Lf, device=dev)
L_ = nat.lib()
w = (torch.randn(L_.alive_filter_block64_weights(), device=dev) * 0.05).to(torch.bfloat16); b = torch.randn(7, 64, device=dev) * 0.1
st = torch.cuda.current_stream().cuda_stream
def run(): nat.check(L_.alive_filter_block64(x.data_ptr(), N, L, w.data_ptr(), b.data_ptr(), film.data_ptr(), 4128, Lf, 3072, skip.data_ptr(), out.data_ptr(), st))
for _ in range(3): run()
nb = N * ((L + 199) // 200)
stamps = torch.zeros(nb, 32, dtype=torch.int64, device=dev)
L_.alive_debug_set_stamps64.argtypes = [C.c_void_p]
L_.alive_debug_set_stamps64(stamps.data_ptr()); run(); torch.cuda.synchronize(); L_.alive_debug_set_stamps64(None)
s = stamps.cpu().double() / 100.0
s = s[s[:, 0] > 0]
names = ["staging", "input conv", "c1 d1", "c2 d1", "c1 d2", "c2 d2", "c1 d4", "c2 d4", "drain + store"]
for wv, off in (("wave 0", 0), ("wave 2", 16)):
    print(wv, "mean us per block:", ", ".join(f"{n} {s[:, off + i].mean().item():.2f}" for i, n in enumerate(names) if n != "-"), " total %.1f" % s[:, off:off + 9].sum(1).mean().item())

