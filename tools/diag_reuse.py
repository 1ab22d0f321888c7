"""which front-end stage is not position-invariant at streaming sizes?  full ring (78 frames) vs slices (30 / 33 frames)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import synthetic
from module.content_encoder import ContentEncoder
from module.f0_estimator import F0Estimator
from module.spectrogram import spectrogram
dev = "cuda"
ce, pe = ContentEncoder(seed=2).to(dev), F0Estimator(seed=2).to(dev)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 78
x = (0.3 * synthetic.make_waveform(F * 320, 64)).to(dev)
spec = spectrogram(x)
for name, a, b in (("left", 0, 32), ("right", F - 35, F), ("middle", 20, 60)):
    s2 = spectrogram(x[:, a * 320:b * 320].contiguous())
    lo, hi = (0 if a == 0 else 2), (b - a if b == F else b - a - 2)
    same = torch.equal(s2[:, :, lo:hi], spec[:, :, a + lo:a + hi])
    print(f"[{name}] spectrogram frames [{a + lo}, {a + hi}) equal: {same}; max diff {float((s2[:, :, lo:hi] - spec[:, :, a + lo:a + hi]).abs().max()):.3e}")
    # networks on the SAME spec values (slice of the full spectrogram), so only their own invariance is tested
    sp = spec[:, :, a:b].contiguous()
    c_full, c_sl = ce(spec), ce(sp)
    p_full, p_sl = pe.estimate(spec), pe.estimate(sp)
    va = 0 if a == 0 else 12
    vb = (b - a) if b == F else (b - a - 12)
    print(f"    content encoder frames [{a + va}, {a + vb}) equal: {torch.equal(c_sl[:, :, va:vb], c_full[:, :, a + va:a + vb])}; max diff "
          f"{float((c_sl[:, :, va:vb] - c_full[:, :, a + va:a + vb]).abs().max()):.3e};  f0 equal: {torch.equal(p_sl[:, :, va:vb], p_full[:, :, a + va:a + vb])}")
