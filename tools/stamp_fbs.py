"""phase stamps of the fused 16- / 8-channel FilterBlocks (diagnostic build: tools/ab_build.sh tools/_ab/fbs_stamps.so filter_small.hip -DALIVE_STAMPS;
ALIVE_VC_LIB=tools/_ab/fbs_stamps.so python tools/stamp_fbs.py)"""
import sys, os, torch, ctypes as Ct
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
dev = "cuda"; N, Lf = 64, 450
st = torch.cuda.current_stream().cuda_stream
f = nat.lib().alive_debug_set_stamps_small; f.argtypes = [Ct.c_void_p]; f.restype = None
for C in (16, 8):
    L = 72000 if C == 16 else 144000
    U = torch.randn(N, C, L, device=dev); skip = torch.randn(N, C, L, device=dev); out = torch.empty_like(U)
    film = torch.randn(N, 12 * C, Lf, device=dev)
    nw = nat.lib().alive_filter_block_small_weights(C)
    w = torch.cat([torch.randn(224, device=dev) * 0.1, (torch.randn(2 * (nw - 224), device=dev) * 0.1).to(torch.bfloat16).view(torch.int16).view(torch.float32)]).contiguous()
    def run(): nat.check(nat.lib().alive_filter_block_small(U.data_ptr(), N, C, L, w.data_ptr(), film.data_ptr(), 12 * C, Lf, 0, skip.data_ptr(), out.data_ptr(), st))
    run(); run()
    TT = (1024 if C == 16 else 2048) - 56
    nb = N * ((L + TT - 1) // TT)
    stamps = torch.zeros(nb, 16, dtype=torch.int64, device=dev)
    f(stamps.data_ptr()); run(); torch.cuda.synchronize(); f(None)
    s = stamps.cpu().double() / 100.0
    s = s[s[:, 0] > 0]
    names = ["prologue", "input conv + z0", "c1 d1", "c2 d1", "c1 d2", "c2 d2", "c1 d4", "c2 d4", "drain + store"]
    print(f"C {C} mean us per block:", ", ".join(f"{n} {s[:, i].mean().item():.2f}" for i, n in enumerate(names)), " total %.1f" % s[:, :9].sum(1).mean().item())
