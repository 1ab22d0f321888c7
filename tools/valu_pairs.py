"""Does a second wave per SIMD double the vector issue rate?  Per-wave shader cycles per slot (slot = one v_mfma_f32_32x32x16_bf16 or
none, + NF independent v_fma_f32) at one wave per SIMD (256 threads per block, one block per CU) and at two (512), csrc/diag.hip."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
L_ = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libalive_diag.so"))
fn = L_.alive_debug_valu_pairs
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
sink = torch.zeros(4, device=dev); cyc = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
rnd = torch.randn(32768, device=dev).to(torch.bfloat16)
iters = 20000
for mf, name in ((0, "no MFMA"), (1, "one MFMA per slot"), (2, "one MFMA per slot, every 8th filler a v_exp_f32")):
    for nf in (0, 4, 8, 16, 24):
        if (mf == 0 and nf in (0, 24)) or (mf == 2 and nf in (0, 4)): continue
        row = []
        for threads in (256, 512):
            cyc.zero_()
            nat.check(fn(rnd.data_ptr(), 256, threads, 200, nf, mf, cyc.data_ptr(), sink.data_ptr(), st))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); nat.check(fn(rnd.data_ptr(), 256, threads, iters, nf, mf, cyc.data_ptr(), sink.data_ptr(), st)); b.record(); torch.cuda.synchronize()
            c = cyc.view(256, 8)[:, : threads // 64].double().median().item() / (iters * 8)
            row.append(f"{threads // 256} wave(s)/SIMD: {c:.1f} cy/slot/wave, wall {a.elapsed_time(b) * 1e6 / (iters * 8):.1f} ns/slot")
        print(f"{name}, {nf} fillers: " + " | ".join(row), flush=True)
