import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo/alive-vc_amd"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
from module import synthetic
from module.common import PackedLibrary
d = np.load("/root/repo/tests/golden/knn_e.npz")
T, M = int(d["T"]), int(d["M"])
lib = synthetic.make_library(M, 12)
base = synthetic.gaussian("knn.base", 13, (1, 768, 1))
lib = base + 0.35 * lib
src = base + 0.35 * synthetic.gaussian("knn.src.e", 11, (1, 768, T))
dev = "cuda"
l16, l8 = PackedLibrary(lib[0].to(dev), prefilter="bf16"), PackedLibrary(lib[0].to(dev), prefilter="fp8")
v0, i0 = l16.search(src.to(dev), 4); v1, i1 = l8.search(src.to(dev), 4)
bad = (~(i0 == i1).all(1)).nonzero()[:, 0].tolist()
print("differing frames", bad, "re-searched", l8.fallback_frames())
s = src[0].t().to(dev); s = s / s.norm(dim=1, keepdim=True)
r = lib[0].t().to(dev); r = r / r.norm(dim=1, keepdim=True)
cos = s @ r.t()
s8 = (s.bfloat16().float() * 256).to(torch.float8_e4m3fn).float(); r8 = (r.bfloat16().float() * 256).to(torch.float8_e4m3fn).float()
cos8 = (s8 @ r8.t()) / 65536
err = (cos8 - cos)
print("fp8 score error: mean %.5f sd %.5f max %.5f" % (err.mean().item(), err.std().item(), err.abs().max().item()))
print("row-to-row sd of the error within a frame: %.5f" % err.std(dim=1).mean().item())
for f in bad:
    print(f, v0[f].tolist(), i0[f].tolist(), "|", v1[f].tolist(), i1[f].tolist())
    rank8 = (cos8[f] > cos8[f][i0[f].long()].unsqueeze(1)).sum(1)
    print("   fp8 ranks of the true neighbours:", rank8.tolist(), " tile/rowclass:", [(int(x) // 32, (int(x) % 8) // 4) for x in i0[f]])
