"""what bf16 MFMA rate the chip sustains on random data at the scoring kernel's occupancy (csrc/diag.hip)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
L_ = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libalive_diag.so"))    # built by csrc/Makefile next to this file
fn = L_.alive_debug_mfma_rate
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
sink = torch.zeros(4, device=dev)
iters = 100000
for name, rnd in (("random", torch.randn(32768, device=dev).to(torch.bfloat16)), ("zeros", torch.zeros(32768, device=dev, dtype=torch.bfloat16))):
    for mode in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
        nat.check(fn(rnd.data_ptr(), 256, 2000, mode, sink.data_ptr(), st))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        nat.check(fn(rnd.data_ptr(), 256, iters, mode, sink.data_ptr(), st))
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        per_iter = 12.0 if mode in (7, 8, 9) else 8.0
        flops = 256 * 4 * per_iter * iters * 2 * 32 * 32 * (16 if mode in (0, 1, 10, 11) else 64)
        what = ["bf16 32x32x16, registers only", "bf16, one ds_read_b128 per two MFMAs", "scaled fp8 32x32x64, registers only",
                "scaled fp6, registers only", "scaled fp4, registers only", "scaled fp8, 32 B of LDS per two MFMAs",
                "scaled fp6, 24 B of LDS per two MFMAs",
                "fp8 A fragment from LDS per TWO MFMAs, fp8 B in registers (knn_score8_kernel today)",
                "fp8 A fragment from LDS per THREE MFMAs, fp6 B in registers (96 stationary frames per wave)",
                "fp8 A fragment from LDS per THREE MFMAs, fp8 B in registers",
                "bf16, one ds_read_b128 per TWO MFMAs, two chains (knn_score_kernel today)",
                "bf16, one ds_read_b128 per ONE MFMA, two chains (32 stationary frames per wave)"][mode]
        print(f"{name:7s} mode {mode} ({what}): {ms:8.1f} ms  {flops / ms / 1e9:7.1f} TFLOP/s")
