"""micro-benchmark of the fused small-channel FilterBlock kernel: python tools/bench_filter_small.py C"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N, Lf = 64, 450
L = 72000 if C == 16 else 144000
dev = "cuda"
U = torch.randn(N, C, L, device=dev); skip = torch.randn(N, C, L, device=dev); out = torch.empty_like(U)
film = torch.randn(N, 12 * C, Lf, device=dev)
w = torch.randn(nat.lib().alive_filter_block_small_weights(C), device=dev) * 0.1
st = torch.cuda.current_stream().cuda_stream
def run():
    nat.check(nat.lib().alive_filter_block_small(U.data_ptr(), N, C, L, w.data_ptr(), film.data_ptr(), 12 * C, Lf, 0, skip.data_ptr(), out.data_ptr(), st))
run(); run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): run()
b.record(); torch.cuda.synchronize()
print(f"C {C}: {a.elapsed_time(b)/5:.3f} ms per call")
if hasattr(nat.lib(), "alive_debug_set_stamps_small"):          # diagnostic build: make -C alive-vc_amd/csrc clean all EXTRA=-DALIVE_STAMPS
    import ctypes as Ct
    TT = 968
    nb = N * ((L + TT - 1) // TT)
    stamps = torch.zeros(nb, 4, dtype=torch.int64, device=dev)
    f = nat.lib().alive_debug_set_stamps_small; f.argtypes = [Ct.c_void_p]; f.restype = None
    f(stamps.data_ptr()); run(); torch.cuda.synchronize(); f(None)
    s = stamps.cpu().double() / 100.0
    print("per tile mean us: FiLM/coords/staging %.1f  input conv %.1f  six convs %.1f  store %.1f" % tuple(s[:, i].mean().item() for i in range(4)))
