"""fused small FilterBlock: the base + immediate form against the per-lane (FIRST) form on the same inputs (ALIVE_FBS_FORCE=1 / 2)"""
import os, sys, subprocess, torch
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(HERE, "..", "alive-vc_amd"))
    from module import _native as nat
    C, N, L, path = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    dev = "cuda"; Lf = 450
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(N, C, L, device=dev, generator=g); skip = torch.randn(N, C, L, device=dev, generator=g); out = torch.empty_like(x)
    film = torch.randn(N, 4128, Lf, device=dev, generator=g)
    L_ = nat.lib()
    nw = L_.alive_filter_block_small_weights(C)
    wb = (torch.randn(2 * (nw - 224), device=dev, generator=g) * 0.1).to(torch.bfloat16)
    w = torch.cat([torch.randn(224, device=dev, generator=g) * 0.1, wb.view(torch.int16).view(torch.float32)]).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for _ in range(2):
        out.zero_()
        nat.check(L_.alive_filter_block_small(x.data_ptr(), N, C, L, w.data_ptr(), film.data_ptr(), 4128, Lf, 100, skip.data_ptr(), out.data_ptr(), st))
        torch.cuda.synchronize(); outs.append(out.cpu().clone())
    torch.save(outs, path); sys.exit(0)
C = int(sys.argv[1]); N = int(sys.argv[2]); L = int(sys.argv[3])
res = {}
for f in ("1", "2"):
    p = f"/tmp/fbs_{f}.pt"
    subprocess.check_call([sys.executable, __file__, "--child", str(C), str(N), str(L), p], env=dict(os.environ, ALIVE_FBS_FORCE=f))
    res[f] = torch.load(p)
a, b = res["1"][0], res["2"][0]
print("finite: first-form", bool(torch.isfinite(a).all()), " base-form", bool(torch.isfinite(b).all()), " run-to-run equal:", torch.equal(res["2"][0], res["2"][1]), torch.equal(res["1"][0], res["1"][1]))
bad = ~torch.isfinite(b) | ((a - b).abs() > 1e-3)
print("elements differing:", int(bad.sum()), "of", bad.numel())
if bad.any():
    n, c, t = bad.nonzero(as_tuple=True)
    TT = (1024 if C == 16 else 2048) - 56
    print("channels", sorted(set(c.tolist())), " tiles", sorted(set((t // TT).tolist()))[:20])
    pos = torch.bincount((t % TT) // 32)
    print("column-in-tile / 32 histogram:", [(i, int(v)) for i, v in enumerate(pos.tolist()) if v][:40])
