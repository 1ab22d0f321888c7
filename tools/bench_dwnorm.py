"""dw conv + ChannelNorm: fp32 kernel + alive_to_planes vs the single-pass plane-packed kernel (128 windows x 450 frames)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
dev = "cuda"; N, T = 128, 450
L_ = nat.lib(); st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps
for C, planes in ((512, 2), (512, 3), (256, 3)):
    x = torch.randn(N, C, T, device=dev); y = torch.empty_like(x)
    w = torch.randn(C, 7, device=dev); b = torch.randn(C, device=dev); g = torch.randn(C, device=dev); o = torch.randn(C, device=dev)
    P = torch.empty(L_.alive_planes_bytes(N * T, C, planes), dtype=torch.uint8, device=dev)
    t1 = timeit(lambda: L_.alive_dwconv_norm(x.data_ptr(), N, C, T, w.data_ptr(), b.data_ptr(), 0, g.data_ptr(), o.data_ptr(), None, 0, 0, 0, 1e-4, y.data_ptr(), st))
    t2 = timeit(lambda: L_.alive_to_planes(y.data_ptr(), N, C, T, planes, P.data_ptr(), st))
    t3 = timeit(lambda: L_.alive_dwconv_norm_planes(x.data_ptr(), N, C, T, w.data_ptr(), b.data_ptr(), 0, g.data_ptr(), o.data_ptr(), None, 0, 0, 0, 1e-4, planes, P.data_ptr(), st))
    print(f"C {C} planes {planes}: fp32 kernel {t1:.3f} + to_planes {t2:.3f} = {t1 + t2:.3f} ms   single pass {t3:.3f} ms")
