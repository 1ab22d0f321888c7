"""three launches of the fused 64-channel FilterBlock at the bench shape (for tools/pmc_one.sh)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
dev = "cuda"; N, L, Lf = 128, 36000, 450
x = torch.randn(N, 64, L, device=dev); skip = torch.randn(N, 64, L, device=dev); out = torch.empty_like(x)
film = torch.randn(N, 4128, Lf, device=dev)
L_ = nat.lib()
w = (torch.randn(L_.alive_filter_block64_weights(), device=dev) * 0.05).to(torch.bfloat16); b = torch.randn(7, 64, device=dev) * 0.1
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    nat.check(L_.alive_filter_block64(x.data_ptr(), N, L, w.data_ptr(), b.data_ptr(), film.data_ptr(), 4128, Lf, 3072, skip.data_ptr(), out.data_ptr(), st))
torch.cuda.synchronize()
