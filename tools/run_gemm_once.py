"""a few launches of one plane-GEMM shape, for counter passes: python tools/run_gemm_once.py ci co planes [reps]"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
from module._pack import pack_conv_split
ci, co, planes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
N, T = 128, 450
L_ = nat.lib(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
x = torch.randn(N, ci, T, device="cuda"); w = torch.randn(co, ci, 1, device="cuda") / ci ** 0.5; b = torch.randn(co, device="cuda")
W = pack_conv_split(w, planes); y = torch.empty(N, co, T, device="cuda")
P = torch.empty(L_.alive_planes_bytes(N * T, ci, planes), dtype=torch.uint8, device="cuda")
nat.check(L_.alive_to_planes(x.data_ptr(), N, ci, T, planes, P.data_ptr(), st))
gd = nat.AliveGemm()
gd.W, gd.bias, gd.P, gd.Y = W.data_ptr(), b.data_ptr(), P.data_ptr(), y.data_ptr()
gd.N, gd.T, gd.Ci, gd.Co, gd.planes, gd.act = N, T, ci, co, planes, 0
for _ in range(reps): nat.check(L_.alive_gemm_planes(C.byref(gd), st))
torch.cuda.synchronize()
print("digest %08x" % (int(y.view(torch.int32).to(torch.int64).sum().item()) & 0xffffffff))
