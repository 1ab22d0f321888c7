"""micro-benchmark of the fused small-channel FilterBlock kernels at the bench shape (64 windows): python tools/bench_filter_small.py [C]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
dev = "cuda"; N, Lf = 64, 450
st = torch.cuda.current_stream().cuda_stream
for C in ([int(sys.argv[1])] if len(sys.argv) > 1 else [16, 8]):
    L = 72000 if C == 16 else 144000
    U = torch.randn(N, C, L, device=dev); skip = torch.randn(N, C, L, device=dev); out = torch.empty_like(U)
    film = torch.randn(N, 12 * C, Lf, device=dev)
    nw = nat.lib().alive_filter_block_small_weights(C)
    w = torch.cat([torch.randn(224, device=dev) * 0.1,        # fp32 biases [7][32], then bf16 weight pairs in fp32 words
                   (torch.randn(2 * (nw - 224), device=dev) * 0.1).to(torch.bfloat16).view(torch.int16).view(torch.float32)]).contiguous()
    def run():
        nat.check(nat.lib().alive_filter_block_small(U.data_ptr(), N, C, L, w.data_ptr(), film.data_ptr(), 12 * C, Lf, 0, skip.data_ptr(), out.data_ptr(), st))
    run(); run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): run()
    b.record(); torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    print(f"C {C}: {a.elapsed_time(b)/5:.3f} ms per {N} windows")
