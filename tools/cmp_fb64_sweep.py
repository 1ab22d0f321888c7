"""the sweep form of the fused 64-channel FilterBlock (fp16 form; csrc/filter_mid.hip SWEEP) against the tiled form: same bits, and the times.
One subprocess per form (ALIVE_FB64_SWEEP is read once per process).  usage: python tools/cmp_fb64_sweep.py [N] [L]"""
import hashlib
import os
import subprocess
import sys

CHILD = r'''
import hashlib, os, sys, torch
sys.path.insert(0, os.path.join(%(root)r, "alive-vc_amd"))
from module import _native as nat
N, L = %(n)d, %(l)d
Lf = L // 80
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(5)
x = 0.3 * torch.randn(N, 64, L, device=dev, generator=g); skip = 0.3 * torch.randn(N, 64, L, device=dev, generator=g); out = torch.empty_like(x)
film = 0.05 * torch.randn(N, 4128, Lf, device=dev, generator=g)          # (gentle operands: a saturating fp16 conversion takes the counted, atomic path)
L_ = nat.lib()
w = (torch.randn(L_.alive_filter_block64_weights(), device=dev, generator=g) * 0.05).to(torch.bfloat16); b = torch.randn(7, 64, device=dev, generator=g) * 0.1
st = torch.cuda.current_stream().cuda_stream
def run(): nat.check(L_.alive_filter_block64_range_fp16(x.data_ptr(), N, L, w.data_ptr(), b.data_ptr(), film.data_ptr(), 4128, Lf, 3072, 0, 0, Lf, skip.data_ptr(), out.data_ptr(), st))
for _ in range(3): run()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
print("RESULT", a.elapsed_time(e) / 10, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest(), bool(torch.isfinite(out).all()), L_.alive_f16_saturations(1))
'''
root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
l = int(sys.argv[2]) if len(sys.argv) > 2 else 36000
res = {}
for sweep in ("0", "1"):
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": root, "n": n, "l": l}], env=dict(os.environ, ALIVE_FB64_SWEEP=sweep),
                       capture_output=True, text=True, timeout=900)
    line = [x for x in r.stdout.splitlines() if x.startswith("RESULT")]
    if not line:
        print(r.stdout[-1500:], r.stderr[-1500:])
        raise SystemExit(1)
    res[sweep] = line[0].split()
    print(f"ALIVE_FB64_SWEEP={sweep}: {float(res[sweep][1]):.3f} ms per {n} windows x {l} columns, digest {res[sweep][2][:16]}, finite {res[sweep][3]}, fp16 saturations {res[sweep][4]}")
print("bitwise equal:", res["0"][2] == res["1"][2])
