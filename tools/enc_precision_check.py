"""content-encoder / f0-estimator outputs against the reference fixtures in both encoder precision modes (alive_encoder_precision):
python tools/enc_precision_check.py   -> relative RMS error of the content features, f0 classes that differ (all / outside the 1e-4 margin)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "alive-vc_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import alive_oracle as O
from module import ops
from module.content_encoder import ContentEncoder
from module.f0_estimator import F0Estimator
dev = "cuda"
ce, pe = ContentEncoder(seed=2).to(dev), F0Estimator(seed=2).to(dev)
for T in (24, 450):
    z = np.load(os.path.join(ROOT, "tests", "golden", f"full_T{T}.npz"))
    wav = torch.from_numpy(z["wav"])
    spec = torch.from_numpy(z["spec"]) if "spec" in z.files else O.spectrogram(wav)
    spec = spec.repeat(8, 1, 1)                      # 8 copies: enough columns for the batch (plane) path at T = 24
    ref = torch.from_numpy(z["feat"]).double()
    for mode in (1, 2):
        ops.encoder_precision(mode)
        feat = ce(spec.to(dev))[:1].cpu().double()
        f = feat if ref.shape[1] == 768 else feat[:, ::8, :]
        f0 = pe.estimate(spec.to(dev))[:1].cpu()
        diff = f0[0, 0] != torch.from_numpy(z["f0"])[0, 0]
        safe = torch.from_numpy(z["f0_margin"])[0] > 1e-4
        print(f"T={T} mode {mode}: content features rel rms error {float((f - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}  "
              f"max abs {float((f - ref).abs().max()):.3e}   f0 classes differing {int(diff.sum())} of {diff.numel()} ({int((diff & safe).sum())} outside the margin)")
ops.encoder_precision(1)
