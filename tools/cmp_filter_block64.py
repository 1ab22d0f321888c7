"""sha256 of the fused 64-channel FilterBlock's output on fixed inputs (first and interior tiles, ragged end) -- run once per build /
per ALIVE_FB64_COMB setting and compare: the column-tile layouts must agree bit for bit.  python tools/cmp_filter_block64.py [N]"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda"; L_ = nat.lib(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(3)
for L, Lf in ((36000, 450), (1000, 13), (404, 6)):
    film = torch.randn(N, 4128, Lf, device=dev, generator=g)
    x = torch.randn(N, 64, L, device=dev, generator=g); skip = torch.randn(N, 64, L, device=dev, generator=g)
    w = (torch.randn(L_.alive_filter_block64_weights(), device=dev, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(7, 64, device=dev, generator=g) * 0.1
    out = torch.empty_like(x)
    nat.check(L_.alive_filter_block64(x.data_ptr(), N, L, w.data_ptr(), b.data_ptr(), film.data_ptr(), 4128, Lf, 3072, skip.data_ptr(), out.data_ptr(), st))
    torch.cuda.synchronize()
    print(f"L {L} Lf {Lf}: finite {bool(torch.isfinite(out).all())}  sha256 {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:20]}  sum {float(out.double().sum()):.6f}")
