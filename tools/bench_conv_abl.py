"""ablation: fixed (prologue + epilogue) cost vs per-tap cost of the split conv kernel"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
from module._pack import pack_conv_split
dev = "cuda"
def run(N, Co, Ci, L, KW, with_z, with_res, Lf=450, reps=10):
    x = torch.randn(N, Ci, L, device=dev); w = torch.randn(Co, Ci, KW, device=dev) * 0.05; b = torch.randn(Co, device=dev)
    W = pack_conv_split(w)
    film = torch.randn(N, 2 * Co, Lf, device=dev)
    y = torch.empty(N, Co, L, device=dev); z = torch.empty(N, Co, L, device=dev); res = torch.randn(N, Co, L, device=dev)
    d = nat.AliveConv()
    d.W, d.bias, d.X = W.data_ptr(), b.data_ptr(), x.data_ptr()
    d.N, d.Ci, d.Tin, d.Co, d.K_pad = N, Ci, L, Co, (W.shape[-1] if W.dim() == 2 else W.shape[1] * 32)
    d.KW, d.stride, d.dil, d.pad_left, d.pad_mode, d.Tout, d.up, d.act = KW, 1, 1, KW - 1, 1, L, 1, 0
    d.Y = y.data_ptr()
    if with_res: d.residual = res.data_ptr()
    if with_z:
        d.Z, d.film, d.film_rows, d.Lf, d.film_scale_row, d.film_shift_row = z.data_ptr(), film.data_ptr(), 2 * Co, Lf, 0, Co
    d.precision, d.Ci_pad = 1, (Ci + 31) // 32 * 32
    L_ = nat.lib(); st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): nat.check(L_.alive_conv1d(C.byref(d), st))
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): L_.alive_conv1d(C.byref(d), st)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps
for (z, r) in ((0, 0), (1, 1)):
    t32 = run(64, 256, 32, 4500, 5, z, r); t256 = run(64, 256, 256, 4500, 5, z, r); t512 = run(64, 256, 512, 4500, 5, z, r)
    print(f"Co256 L4500 k5 z={z} res={r}: Ci=32 {t32:.3f} ms, Ci=256 {t256:.3f} ms, Ci=512 {t512:.3f} ms -> per 32-ch block {(t512-t256)/8*1e3:.1f} us, fixed {t32-(t512-t256)/8:.3f} ms")
    t32 = run(64, 64, 32, 36000, 5, z, r); t64 = run(64, 64, 64, 36000, 5, z, r); t128 = run(64, 64, 128, 36000, 5, z, r)
    print(f"Co64 L36000 k5 z={z} res={r}: Ci=32 {t32:.3f} ms, Ci=64 {t64:.3f} ms, Ci=128 {t128:.3f} ms -> per block {(t128-t64)/2*1e3:.1f} us, fixed {t32-(t128-t64)/2:.3f} ms")
