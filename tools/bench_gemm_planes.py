"""micro-benchmark: frame-rate 1x1 convs, conv_split kernel (fp32 activations, split on the fly) vs plane-packed GEMM"""
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


def run(name, ci, co, planes, act=0, res=False, pout=False):
    torch.manual_seed(ci * 7 + co)
    x = torch.randn(N, ci, T, device=dev); w = torch.randn(co, ci, 1, device=dev) / ci ** 0.5; b = torch.randn(co, device=dev)
    W = pack_conv_split(w, planes)
    y = torch.empty(N, co, T, device=dev); r = torch.randn(N, co, T, device=dev)
    d = nat.AliveConv()
    d.W, d.bias, d.X = W.data_ptr(), b.data_ptr(), x.data_ptr()
    d.N, d.Ci, d.Tin, d.Co, d.K_pad = N, ci, T, co, (W.shape[-1] if W.dim() == 2 else W.shape[1] * 32)
    d.KW, d.stride, d.dil, d.pad_left, d.pad_mode, d.Tout, d.up, d.act = 1, 1, 1, 0, 0, T, 1, act
    d.Y = y.data_ptr()
    if res: d.residual = r.data_ptr()
    d.precision, d.Ci_pad = planes - 1, (ci + 31) // 32 * 32
    t_old = timeit(lambda: L_.alive_conv1d(C.byref(d), st))
    P = torch.empty(L_.alive_planes_bytes(N * T, ci, planes), dtype=torch.uint8, device=dev)
    t_cvt = timeit(lambda: L_.alive_to_planes(x.data_ptr(), N, ci, T, planes, P.data_ptr(), st))
    Po = torch.empty(L_.alive_planes_bytes(N * T, co, planes), dtype=torch.uint8, device=dev)
    gd = nat.AliveGemm()
    gd.W, gd.bias, gd.P = W.data_ptr(), b.data_ptr(), P.data_ptr()
    gd.N, gd.T, gd.Ci, gd.Co, gd.planes, gd.act = N, T, ci, co, planes, act
    if res: gd.residual = r.data_ptr()
    if pout: gd.Pout = Po.data_ptr()
    else: gd.Y = y.data_ptr()
    nat.check(L_.alive_gemm_planes(C.byref(gd), st))
    t_new = timeit(lambda: L_.alive_gemm_planes(C.byref(gd), st))
    fl = 2.0 * ci * co * N * T * (3 if planes == 2 else 6)
    out = Po.view(torch.int16) if pout else y.view(torch.int32)
    digest = int(out.to(torch.int64).sum().item()) & 0xffffffff          # (seeded inputs: equal digests across builds = equal bits)
    print(f"{name:26s} planes={planes} act={act} res={int(res)} pout={int(pout)}  split {t_old:6.3f} ms ({fl / t_old / 1e9:6.0f} TF)   "
          f"gemm_planes {t_new:6.3f} ms ({fl / t_new / 1e9:6.0f} TF)   to_planes {t_cvt:6.3f} ms   digest {digest:08x}", flush=True)


for planes in (2, 3):
    run("in 641->512", 641, 512, planes)
    run("pw1 512->1536 gelu", 512, 1536, planes, act=1)
    run("pw1 512->1536 gelu pout", 512, 1536, planes, act=1, pout=True)
    run("pw2 1536->512 res", 1536, 512, planes, res=True)
    run("out 512->768", 512, 768, planes)
    run("film 512->4128", 512, 4128, planes)
    run("pe pw1 256->512 gelu", 256, 512, planes, act=1, pout=True)
    run("pe pw2 512->256 res", 512, 256, planes, res=True)
    run("pe out 256->4096", 256, 4096, planes)
