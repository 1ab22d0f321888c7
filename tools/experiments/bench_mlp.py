"""time of the fused ConvNeXt pointwise pair alone (csrc/mlp_fused.hip): python tools/bench_mlp.py [windows] [frames]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat, ops
from module._pack import pack_conv_split_h
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 450
C, H = 512, 1536
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
L_ = nat.lib()
cols = N * T; cols_pad = (cols + 127) // 128 * 128
P = (torch.randn(C // 32, cols_pad, 32, device=dev, generator=g)).to(torch.float16)
w1 = torch.randn(H, C, 1, device=dev, generator=g) * (1.0 / C ** 0.5); w2 = torch.randn(C, H, 1, device=dev, generator=g) * (1.0 / H ** 0.5)
W1 = pack_conv_split_h(w1)[2].contiguous(); W2 = pack_conv_split_h(w2)[2].contiguous()
b1 = torch.randn(H, device=dev, generator=g) * 0.1; b2 = torch.randn(C, device=dev, generator=g) * 0.1; sc = torch.rand(C, device=dev, generator=g)
x = torch.randn(N, C, T, device=dev, generator=g)
st = torch.cuda.current_stream().cuda_stream
def run(): nat.check(L_.alive_convnext_mlp_fp16(P.data_ptr(), N, T, C, H, W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), sc.data_ptr(), x.data_ptr(), st))
for _ in range(2): run()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): run()
e.record(); torch.cuda.synchronize()
ms = a.elapsed_time(e) / 5
print(f"fused: {ms:.3f} ms per {N} x {T} frames; {2 * 2 * C * H * cols / ms / 1e9:.0f} TFLOP/s; finite {bool(torch.isfinite(x).all())} sat {L_.alive_f16_saturations(1)}")
