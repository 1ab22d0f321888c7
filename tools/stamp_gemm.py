"""phase timing of gemm_planes blocks from in-kernel wall-clock stamps (100 MHz counter).
Needs the diagnostic build: make -C alive-vc_amd/csrc clean all EXTRA=-DALIVE_STAMPS (the production kernels carry no stamps)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
from module._pack import pack_conv_split
dev = "cuda"; N, T = 128, 450
L_ = nat.lib(); st = torch.cuda.current_stream().cuda_stream
L_.alive_debug_set_stamps.argtypes = [C.c_void_p]
def run(ci, co, planes, act, res, pout):
    x = torch.randn(N, ci, T, device=dev); w = torch.randn(co, ci, 1, device=dev) / ci ** 0.5; b = torch.randn(co, device=dev)
    W = pack_conv_split(w, planes); y = torch.empty(N, co, T, device=dev); r = torch.randn(N, co, T, device=dev)
    P = torch.empty(L_.alive_planes_bytes(N * T, ci, planes), dtype=torch.uint8, device=dev)
    L_.alive_to_planes(x.data_ptr(), N, ci, T, planes, P.data_ptr(), st)
    Po = torch.empty(L_.alive_planes_bytes(N * T, co, planes), dtype=torch.uint8, device=dev)
    gd = nat.AliveGemm(); gd.W, gd.bias, gd.P = W.data_ptr(), b.data_ptr(), P.data_ptr()
    gd.N, gd.T, gd.Ci, gd.Co, gd.planes, gd.act = N, T, ci, co, planes, act
    if res: gd.residual = r.data_ptr()
    if pout: gd.Pout = Po.data_ptr()
    else: gd.Y = y.data_ptr()
    nb = ((co + 127) // 128) * ((N * T + 127) // 128)
    stamps = torch.zeros(nb, 8, dtype=torch.int64, device=dev)
    for _ in range(3): L_.alive_gemm_planes(C.byref(gd), st)
    L_.alive_debug_set_stamps(stamps.data_ptr())
    nat.check(L_.alive_gemm_planes(C.byref(gd), st)); torch.cuda.synchronize()
    L_.alive_debug_set_stamps(None)
    s = stamps.cpu().double()
    t0 = s[:, 0].min()
    d = (s[:, 1:5] - s[:, 0:4]) / 100.0      # us
    print(f"ci {ci} co {co} planes {planes} act {act} res {res} pout {pout}: blocks {nb}  total {(s[:,4].max()-t0)/100:.1f} us | "
          f"per block mean us: prologue {d[:,0].mean():.2f}  first-stage wait {d[:,1].mean():.2f}  loop {d[:,2].mean():.2f}  epilogue {d[:,3].mean():.2f} | "
          f"block total {((s[:,4]-s[:,0])/100).mean():.2f} | loop cycles {s[:,5].mean():.0f} ({s[:,5].mean() / d[:,2].mean() / 1e3:.2f} GHz): "
          f"vmcnt wait {100 * s[:,6].mean() / s[:,5].mean():.1f} %  barrier wait {100 * s[:,7].mean() / s[:,5].mean():.1f} %")
for planes in (2, 3):
    run(512, 1536, planes, 1, 0, 0); run(512, 1536, planes, 1, 0, 1); run(1536, 512, planes, 0, 1, 0); run(256, 4096, planes, 0, 0, 0)
