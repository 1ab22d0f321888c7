"""A/B of the 256-channel filter convs (conv_split_kernel): python tools/bench_conv256.py  (ALIVE_CONV_TILE256=0|1 selects the tile)
also checks the two tiles against each other (bitwise: same k order per output element)"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
from module._pack import pack_conv_split
dev = "cuda"
N, C_, L, Lf = 128, 256, 4500, 450
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(N, C_, L, device=dev, generator=g); res = torch.randn(N, C_, L, device=dev, generator=g)
film = torch.randn(N, 2 * C_, Lf, device=dev, generator=g)
L_ = nat.lib(); st = torch.cuda.current_stream().cuda_stream
out = {}
for KW, dil, with_z, with_res, want_y in ((5, 2, 1, 1, 1), (5, 4, 1, 0, 0), (1, 1, 1, 0, 1), (5, 1, 0, 1, 1)):
    w = torch.randn(C_, C_, KW, device=dev, generator=g) * 0.05; b = torch.randn(C_, device=dev, generator=g)
    W = pack_conv_split(w)
    y = torch.empty(N, C_, L, device=dev); z = torch.empty(N, C_, L, device=dev)
    d = nat.AliveConv()
    d.W, d.bias, d.X = W.data_ptr(), b.data_ptr(), x.data_ptr()
    d.N, d.Ci, d.Tin, d.Co, d.K_pad = N, C_, L, C_, (W.shape[-1] if W.dim() == 2 else W.shape[1] * 32)
    d.KW, d.stride, d.dil, d.pad_left, d.pad_mode, d.Tout, d.up, d.act = KW, 1, dil, (KW - 1) * dil, 1, L, 1, 0
    if want_y: d.Y = y.data_ptr()
    if with_res: d.residual = res.data_ptr()
    if with_z: d.Z, d.film, d.film_rows, d.Lf, d.film_scale_row, d.film_shift_row = z.data_ptr(), film.data_ptr(), 2 * C_, Lf, 0, C_
    d.precision, d.Ci_pad = 1, C_
    for _ in range(2): nat.check(L_.alive_conv1d(C.byref(d), st))
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): L_.alive_conv1d(C.byref(d), st)
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 10
    fl = 2.0 * C_ * C_ * KW * L * N
    # the same conv with plane-packed operands (AliveConv.Xp / Zp: LDS-DMA staging, plane second output)
    xp = torch.empty(L_.alive_planes_bytes(N * L, C_, 2), dtype=torch.uint8, device=dev)
    L_.alive_to_planes(x.data_ptr(), N, C_, L, 2, xp.data_ptr(), st)
    zp = torch.empty(L_.alive_planes_bytes(N * L, C_, 2), dtype=torch.uint8, device=dev)
    d.Xp, d.X = xp.data_ptr(), None
    if with_z: d.Zp, d.Z = zp.data_ptr(), None
    for _ in range(2): nat.check(L_.alive_conv1d(C.byref(d), st))
    a.record()
    for _ in range(10): L_.alive_conv1d(C.byref(d), st)
    e.record(); torch.cuda.synchronize()
    ms_p = a.elapsed_time(e) / 10
    # plain fp16 (precision 3): one fp16 plane in / out, one MFMA per product; checked against float64 on the rounded operands
    from module._pack import pack_conv_split_h
    Wh = pack_conv_split_h(w)
    x1 = torch.empty(L_.alive_planes_bytes(N * L, C_, 1), dtype=torch.uint8, device=dev)
    L_.alive_to_planes(x.data_ptr(), N, C_, L, 1, x1.data_ptr(), st)
    d.precision, d.W, d.Xp = 3, Wh[2].data_ptr(), x1.data_ptr()
    for _ in range(2): nat.check(L_.alive_conv1d(C.byref(d), st))
    a.record()
    for _ in range(10): L_.alive_conv1d(C.byref(d), st)
    e.record(); torch.cuda.synchronize()
    ms_b = a.elapsed_time(e) / 10
    if want_y:
        xb, wb = x[:2].half().double(), w.half().double()
        ref = torch.nn.functional.conv1d(torch.nn.functional.pad(xb, ((KW - 1) * dil, 0), mode="reflect") if KW > 1 else xb, wb, b.double(), dilation=dil)
        if with_res: ref = ref + res[:2].double()
        print(f"    fp16: max |y - float64 on rounded operands| {float((y[:2].double() - ref).abs().max()):.3e}")
    d.precision, d.W, d.Xp = 1, W.data_ptr(), xp.data_ptr()
    print(f"k{KW} d{dil} z={with_z} res={with_res} y={want_y}: {ms:7.3f} ms (planes in/out {ms_p:7.3f} ms, plain fp16 {ms_b:7.3f} ms)   {fl / ms / 1e9:7.1f} TF-eq  ({3 * fl / ms / 1e12:.2f} PF bf16)   checksum y {float(y.double().sum()) if want_y else 0:.6e} z {float(z.double().sum()) if with_z else 0:.6e}")
