"""one-plane (plain fp16) plane GEMM forms: ALIVE_GEMM1_FORM=0..3 python tools/bench_gemm1.py   (ms per 128 windows x 450 columns)"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
from module._pack import pack_conv_split
dev = "cuda"
N, T = 128, 450
L_ = nat.lib()
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps
for name, ci, co, act, res, pout in (("pw1 512->1536 gelu pout", 512, 1536, 1, False, True), ("pw2 1536->512 res", 1536, 512, 0, True, False),
                                     ("normfilm 512->4096", 512, 4096, 0, False, False)):
    torch.manual_seed(ci * 7 + co)
    x = torch.randn(N, ci, T, device=dev); w = torch.randn(co, ci, 1, device=dev) / ci ** 0.5; b = torch.randn(co, device=dev)
    from module._pack import pack_conv_split_h
    W3 = pack_conv_split_h(w); W = W3
    y = torch.empty(N, co, T, device=dev); r = torch.randn(N, co, T, device=dev)
    P = torch.empty(L_.alive_planes_bytes(N * T, ci, 2), dtype=torch.uint8, device=dev)
    L_.alive_to_planes(x.data_ptr(), N, ci, T, 2, P.data_ptr(), st)
    P1 = torch.empty(L_.alive_planes_bytes(N * T, ci, 1), dtype=torch.uint8, device=dev)
    L_.alive_to_planes(x.data_ptr(), N, ci, T, 1, P1.data_ptr(), st)
    Po = torch.empty(L_.alive_planes_bytes(N * T, co, 2), dtype=torch.uint8, device=dev)
    out = []
    for planes in (2, 1):
        gd = nat.AliveGemm()
        gd.W, gd.bias, gd.P = (W3[2].data_ptr() if planes == 1 else W.data_ptr()), b.data_ptr(), (P1 if planes == 1 else P).data_ptr()
        gd.N, gd.T, gd.Ci, gd.Co, gd.planes, gd.act = N, T, ci, co, planes, act
        if res: gd.residual = r.data_ptr()
        if pout: gd.Pout = Po.data_ptr()
        else: gd.Y = y.data_ptr()
        nat.check(L_.alive_gemm_planes(C.byref(gd), st))
        out.append(timeit(lambda: L_.alive_gemm_planes(C.byref(gd), st)))
    fl = 2.0 * ci * co * N * T
    print(f"{name:26s} two planes {out[0]:6.3f} ms   one plane {out[1]:6.3f} ms ({fl / out[1] / 1e9:6.0f} TF = {fl / out[1] / 1e9 / 2500:.2f} of the bf16 peak)", flush=True)
