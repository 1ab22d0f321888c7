"""cycles per v_mfma_f32_32x32x16_bf16 with N v_fma_f32 issued behind every MFMA, one wave per SIMD on every CU (csrc/diag.hip)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
L_ = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libalive_diag.so"))
fn = L_.alive_debug_mfma_filler
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
sink = torch.zeros(4, device=dev); cyc = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
rnd = torch.randn(32768, device=dev).to(torch.bfloat16)
iters = 20000
for kind, name in ((0, "dependent MFMA chain, independent fillers"), (1, "dependent MFMA chain, fillers = one dependent chain"), (2, "two alternating accumulators, independent fillers"),
                   (3, "v_fmaak_f32 with a 32-bit literal"), (4, "v_exp_f32"), (5, "v_cvt_pk_bf16_f32"), (6, "ds_read_b128 (same 1 KB per wave)"),
                   (7, "v_accvgpr_read_b32 of another accumulator"), (8, "v_fma_f32 with |src| modifier"), (9, "ds_write_b64"),
                   (10, "v_fma_f32 fillers, MFMA A operand in an AGPR"), (11, "v_fma_f32 fillers + one ds_read_b128 per MFMA"),
                   (12, "v_fma_f32 fillers, accumulator in VGPRs")):
    if len(sys.argv) > 1 and str(kind) not in sys.argv[1:]: continue
    row = []
    for nf in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
        nat.check(fn(rnd.data_ptr(), 256, 200, nf, kind, cyc.data_ptr(), sink.data_ptr(), st))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); nat.check(fn(rnd.data_ptr(), 256, iters, nf, kind, cyc.data_ptr(), sink.data_ptr(), st)); b.record(); torch.cuda.synchronize()
        c = cyc.double().median().item() / (iters * 8)
        row.append(f"{nf}:{c:.1f}cy/{a.elapsed_time(b) * 1e6 / (iters * 8):.1f}ns")
    print(f"{name}: fillers per MFMA : shader cycles per MFMA / wall ns per MFMA\n   " + "  ".join(row))
