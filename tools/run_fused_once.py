"""one seeded launch of a fused FilterBlock, digest of the output: python tools/run_fused_once.py C L N
(the tile form is chosen by ALIVE_FB64_NT / ALIVE_FBS_PLANE: equal digests across forms = equal bits)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import ops
c, l, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(c * 1000 + l)
lf = max(5, l // {64: 80, 16: 160, 8: 320}[c])          # samples per frame at each scale of the Filter
x = torch.randn(n, c, l).cuda(); film = (torch.randn(n, 6 * 2 * c + 5, lf) * 0.3).cuda(); skip = torch.randn(n, c, l).cuda()
sd = {"n.input_conv.weight": torch.randn(c, c, 1) * 0.1, "n.input_conv.bias": torch.randn(c) * 0.1}
for j in range(3):
    for cc in ("c1", "c2"):
        p = f"n.blocks.{j}.{cc}"
        sd[p + ".conv.conv.weight"] = torch.randn(c, c, 5) * (0.4 / (5 * c) ** 0.5)
        sd[p + ".conv.conv.bias"] = torch.randn(c) * 0.1
fused = ops.filter_block64 if c == 64 else ops.filter_block_small
out = fused(x, {k: v.cuda() for k, v in sd.items()}, "n", film, 5, skip=skip)
torch.cuda.synchronize()
assert torch.isfinite(out).all()
print("digest %08x" % (int(out.view(torch.int32).to(torch.int64).sum().item()) & 0xffffffff))
