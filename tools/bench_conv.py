"""micro-benchmark of alive_conv1d shapes (decoder filter layers), fp32 vs split-bf16, with/without the FiLM epilogue"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
from module._pack import pack_conv, pack_conv_split
dev = "cuda"
def run(name, N, C_, L, KW, dil, split, with_z, with_res, Lf=450, reps=10):
    x = torch.randn(N, C_, L, device=dev); w = torch.randn(C_, C_, KW, device=dev) * 0.05; b = torch.randn(C_, device=dev)
    W = pack_conv_split(w) if split else pack_conv(w)
    film = torch.randn(N, 2 * C_, Lf, device=dev)
    y = torch.empty(N, C_, L, device=dev); z = torch.empty(N, C_, L, device=dev); res = torch.randn(N, C_, L, device=dev)
    d = nat.AliveConv()
    d.W, d.bias, d.X = W.data_ptr(), b.data_ptr(), x.data_ptr()
    d.N, d.Ci, d.Tin, d.Co, d.K_pad = N, C_, L, C_, (W.shape[-1] if W.dim() == 2 else W.shape[1] * 32)
    d.KW, d.stride, d.dil, d.pad_left, d.pad_mode, d.Tout, d.up, d.act = KW, 1, dil, (KW - 1) * dil, 1, L, 1, 0
    d.Y = y.data_ptr()
    if with_res: d.residual = res.data_ptr()
    if with_z:
        d.Z, d.film, d.film_rows, d.Lf, d.film_scale_row, d.film_shift_row = z.data_ptr(), film.data_ptr(), 2 * C_, Lf, 0, C_
    d.precision, d.Ci_pad = (1, (C_ + 31) // 32 * 32) if split else (0, 0)
    L_ = nat.lib()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): nat.check(L_.alive_conv1d(C.byref(d), st))
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): L_.alive_conv1d(C.byref(d), st)
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / reps
    fl = 2.0 * C_ * C_ * KW * L * N
    print(f"{name:28s} split={int(split)} z={int(with_z)} res={int(with_res)}  {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF-eq")
for split in (0, 1):
    for z, r in ((0, 0), (0, 1), (1, 1)):
        run("scale0 C256 L4500 k5", 64, 256, 4500, 5, 2, split, z, r)
for split in (0, 1):
    for z, r in ((0, 0), (1, 1)):
        run("scale1 C64 L36000 k5", 64, 64, 36000, 5, 2, split, z, r)
for z, r in ((0, 0), (1, 1)):
    run("scale2 C16 L72000 k5", 64, 16, 72000, 5, 2, 0, z, r)
    run("scale3 C8 L144000 k5", 64, 8, 144000, 5, 2, 0, z, r)
run("pw 512->512 T450 k1", 64, 512, 450, 1, 1, 1, 0, 0)
run("pw 512->512 T450 k1", 64, 512, 450, 1, 1, 0, 0, 0)
