"""Run-to-run determinism of the fused FilterBlock kernels at the bench shape (128 windows): `reps` launches on the same
inputs, every output compared bitwise with the first.  ALIVE_VC_LIB selects the library under test (e.g. a build with
-DALIVE_FB64_NO_CHAIN_GAP).   python tools/stress_filter_block.py [reps] [N]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = "cuda"; Lf = 450
L_ = nat.lib()
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(3)
film = torch.randn(N, 4128, Lf, device=dev, generator=g)
print("library:", nat.LIB_PATH)
for C, L in ((64, 36000), (16, 72000), (8, 144000)):
    x = torch.randn(N, C, L, device=dev, generator=g); skip = torch.randn(N, C, L, device=dev, generator=g)
    out = torch.empty_like(x); first = None
    if C == 64:
        w = (torch.randn(L_.alive_filter_block64_weights(), device=dev, generator=g) * 0.05).to(torch.bfloat16)
        b = torch.randn(7, 64, device=dev, generator=g) * 0.1
        run = lambda: nat.check(L_.alive_filter_block64(x.data_ptr(), N, L, w.data_ptr(), b.data_ptr(), film.data_ptr(), 4128, Lf, 3072,
                                                        skip.data_ptr(), out.data_ptr(), st))
    else:
        nw = L_.alive_filter_block_small_weights(C)
        w = (torch.cat([torch.randn(224, device=dev, generator=g) * 0.1,        # fp32 biases [7][32], then bf16 weight pairs in fp32 words
                       (torch.randn(2 * (nw - 224), device=dev, generator=g) * 0.1).to(torch.bfloat16).view(torch.int16).view(torch.float32)]).contiguous())
        run = lambda: nat.check(L_.alive_filter_block_small(x.data_ptr(), N, C, L, w.data_ptr(), film.data_ptr(), 4128, Lf, 100,
                                                            skip.data_ptr(), out.data_ptr(), st))
    bad = 0; worst = 0.0
    for r in range(reps):
        out.zero_()
        run()
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
            assert torch.isfinite(first).all()
        elif not torch.equal(out, first):
            bad += 1
            worst = max(worst, float((out - first).abs().max()))
    print(f"C={C} L={L} N={N}: {reps} launches, {bad} differ from the first (max |diff| {worst:.3e})", flush=True)
