"""The fused 256-channel FilterBlock (csrc/filter_big.hip, round 6) against the conv-by-conv form of round 5 (ALIVE_FB256=0), through the
decoder: waveform RMS error against every reference fixture (tests/golden/full_T450*.npz), the difference between the two forms, fp16
saturations, and the decoder's time on a batch of windows.  One subprocess per form (the switch is read once per process).
usage: python tools/cmp_fb256.py [windows] [out.json] [switch]      switch: ALIVE_FB256 (default) or ALIVE_FB64S (the 64-channel block on
the same kernel against filter_mid.hip's sweep kernel)"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CHILD = r'''
import json, os, sys, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.join(%(root)r, "alive-vc_amd")); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import alive_oracle as O
from module import schema, synthetic, ops
from module.decoder import Decoder
res = {}
for tag in ("", "s3", "s5", "x4"):
    name = "full_T450" + ("_" + tag if tag else "")
    z = np.load(os.path.join(%(root)r, "tests", "golden", name + ".npz"))
    if tag:
        sce, _, sdec = synthetic.fixture_state_dicts(tag)
    else:
        sce = synthetic.make_state_dict(schema.content_encoder_schema(), 2, "ce.")
        sdec = synthetic.make_state_dict(schema.decoder_schema(), 2, "dec.")
    dec = Decoder(); dec.load_state_dict(sdec); dec = dec.to("cuda"); dec.precision = 1
    feat = O.content_encoder(sce, O.spectrogram(torch.from_numpy(z["wav"]))).cuda()
    f0 = torch.from_numpy(z["f0_dec"]).cuda()
    wave, _ = dec(feat, f0)
    ref = torch.from_numpy(z["wave"])
    err = (wave.double().cpu() - ref.double()).pow(2).mean().sqrt().item()
    res[name] = {"rms_error": err, "waveform_rms": ref.double().pow(2).mean().sqrt().item(), "saturations": ops.f16_saturations(reset=True),
                 "finite": bool(torch.isfinite(wave).all())}
    np.save(os.path.join(%(out)r, name + ".npy"), wave.cpu().numpy())
    if not tag:
        n = %(n)d
        fb, f0b = feat.expand(n, -1, -1).contiguous(), f0.expand(n, -1, -1).contiguous() if f0.dim() == 3 else f0.expand(n, -1).contiguous()
        for _ in range(2): wb, _ = dec(fb, f0b)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): wb, _ = dec(fb, f0b)
        e.record(); torch.cuda.synchronize()
        res["decoder_ms_per_%%d_windows" %% n] = a.elapsed_time(e) / 5
        res["batch_rows_equal_the_single_window"] = bool((wb == wave).all())
        ops.f16_saturations(reset=True)
print("RESULT " + json.dumps(res))
'''


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    switch = sys.argv[3] if len(sys.argv) > 3 else "ALIVE_FB256"
    out = {}
    import numpy as np
    for form in ("0", "1"):
        d = os.path.join(ROOT, "gpurun_out", "cmp_fb256_" + form)
        os.makedirs(d, exist_ok=True)
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "n": n, "out": d}], env=dict(os.environ, **{switch: form}),
                           capture_output=True, text=True, timeout=1200)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            print(r.stdout[-3000:], r.stderr[-3000:])
            raise SystemExit(f"{switch}={form} failed")
        out[form] = json.loads(line[0][7:])
        print(f"{switch}={form}: " + json.dumps(out[form]), flush=True)
    diff = {}
    for name in [k for k in out["0"] if k.startswith("full")]:
        a = np.load(os.path.join(ROOT, "gpurun_out", "cmp_fb256_0", name + ".npy")).astype(np.float64)
        b = np.load(os.path.join(ROOT, "gpurun_out", "cmp_fb256_1", name + ".npy")).astype(np.float64)
        diff[name] = float(np.sqrt(((a - b) ** 2).mean()))
    out["rms_between_the_forms"] = diff
    print("rms between the forms:", diff)
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
