"""Where do the results of the -DALIVE_NO_TILE_FENCE build of filter_block64_kernel differ from the shipped build?
Runs both libraries on the same inputs (the shipped one is the truth) and prints the distribution of mismatching elements
over (column tile ct, column in tile c16, wave, kq, e).  python tools/diag_filter_nofence.py <nofence.so> [N]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import _native as nat
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L, Lf, dev = 36000, 450, "cuda"
good = nat.lib()
bad = C.CDLL(sys.argv[1])
for name in ("alive_filter_block64",):
    fn = getattr(bad, name); fn.restype, fn.argtypes = nat.PROTOTYPES[name]
g = torch.Generator(device=dev).manual_seed(3)
film = torch.randn(N, 4128, Lf, device=dev, generator=g)
x = torch.randn(N, 64, L, device=dev, generator=g); skip = torch.zeros(N, 64, L, device=dev)
w = (torch.randn(good.alive_filter_block64_weights(), device=dev, generator=g) * 0.05).to(torch.bfloat16)
b = torch.randn(7, 64, device=dev, generator=g) * 0.1
st = torch.cuda.current_stream().cuda_stream
def run(lib):
    out = torch.empty_like(x)
    rc = lib.alive_filter_block64(x.data_ptr(), N, L, w.data_ptr(), b.data_ptr(), film.data_ptr(), 4128, Lf, 3072, skip.data_ptr(), out.data_ptr(), st)
    assert rc == 0
    torch.cuda.synchronize()
    return out
import itertools
def experiments():
    """which operand carries the error?  (a) as is; (b) FiLM constant over frames (interpolation weights cannot matter);
    (c) identity FiLM; (d) zero conv weights (accumulators = bias: only the epilogue arithmetic is live)"""
    yield "as is", None
    f0 = film.clone(); yield "film constant over frames", lambda: film.copy_(f0[:, :, :1].expand(-1, -1, Lf))
    def ident():
        film.zero_(); film[:, 3072:3072 + 768].view(N, 6, 2, 64, Lf)[:, :, 0] = 1.0
    yield "identity film", ident
    def zw():
        film.copy_(f0); w.zero_()
    yield "zero weights, random film", zw
for name, prep in experiments():
    if prep is not None: prep()
    ref = run(good)
    assert torch.equal(ref, run(good))
    o = run(bad)
    diff = (o != ref)
    ct = ((diff.nonzero(as_tuple=True)[2] % 200 + 56) // 16) if int(diff.sum()) else torch.zeros(0, dtype=torch.long, device=dev)
    print(f"[{name}] {int(diff.sum())} of {o.numel()} differ, max |diff| {float((o - ref).abs().max()):.3e}; by ct {torch.bincount(ct, minlength=16).tolist()}")
sys.exit(0)
for rep in range(3):
    o = run(bad)
    diff = (o != ref)
    nbad = int(diff.sum())
    print(f"rep {rep}: {nbad} of {o.numel()} elements differ, max |diff| {float((o - ref).abs().max()):.3e}")
    if nbad == 0: continue
    n_i, ch, t = diff.nonzero(as_tuple=True)
    local = t % 200 + 56
    ct, c16 = local // 16, local % 16
    wv, kq, e = ch // 16, (ch % 16) // 4, ch % 4
    for name, v, m in (("tile (t // 200)", t // 200, 180), ("ct", ct, 16), ("c16", c16, 16), ("wave", wv, 4), ("kq", kq, 4), ("e", e, 4), ("window", n_i, N)):
        print(f"   by {name:16s}", torch.bincount(v, minlength=m).tolist()[:40])
    # size of the error relative to the value
    rel = ((o - ref).abs() / ref.abs().clamp(min=1e-3))[diff]
    print("   relative error quantiles", [round(float(q), 4) for q in torch.quantile(rel.float()[:1000000], torch.tensor([0.1, 0.5, 0.9, 0.99], device=dev))])
