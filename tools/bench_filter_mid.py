"""time the fused 64-channel FilterBlock (csrc/filter_mid.hip) at the bench shape: 128 windows x 36000 samples"""
import sys, os, torch
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
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
ms = a.elapsed_time(e) / 10
fl = 2.0 * 64 * (64 + 6 * 320) * N * L * 3
print(f"fused FilterBlock C=64: {ms:.3f} ms per 128 windows  ({fl / ms / 1e9:.0f} TF MFMA-equivalent, {3 * x.numel() * 4 / ms / 1e6:.0f} GB/s of tensor I/O)")
print("(phase stamps: tools/ab_build.sh tools/_ab/fb64_stamps.so filter_mid.hip -DALIVE_STAMPS; ALIVE_VC_LIB=tools/_ab/fb64_stamps.so python tools/stamp_fb64.py)")
