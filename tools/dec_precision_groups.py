"""Which layer group of decoder precision mode 1 costs what, on every reference fixture (round 6): the decoder's waveform RMS error against
tests/golden/full_T450*.npz with ONE group on plain fp16 at a time (ALIVE_DECODER_BF16_MASK: (a) = 1 the 256-channel k5 convs, (b) = 2 the
ConvNeXt pointwise convs, (c) = 4 norm-FiLM projection + coarse down convs + mid conv, (e) = 16 the 64-channel block's k5 convs), all of
them (23 = the default of mode 1) and none (0 = every GEMM on split bf16).  The mask is read once per process: one subprocess per mask.
usage: python tools/dec_precision_groups.py [out.json]"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import json, os, sys
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
    dec = Decoder(); dec.load_state_dict(sdec); dec = dec.to("cuda")
    feat = O.content_encoder(sce, O.spectrogram(torch.from_numpy(z["wav"]))).cuda()
    wave, _ = dec(feat, torch.from_numpy(z["f0_dec"]).cuda())
    ref = torch.from_numpy(z["wave"])
    err = (wave.double().cpu() - ref.double()).pow(2).mean().sqrt().item()
    res[name] = {"rms_error": err, "waveform_rms": ref.double().pow(2).mean().sqrt().item(), "saturations": ops.f16_saturations(reset=True)}
print("RESULT " + json.dumps(res))
'''


def main():
    out = {}
    for mask, what in ((0, "none (split bf16 everywhere)"), (1, "(a) 256-channel k5 convs"), (2, "(b) ConvNeXt pointwise convs"),
                       (4, "(c) norm-FiLM projection, coarse down convs, mid conv"), (16, "(e) 64-channel block k5 convs"),
                       (7, "(a) + (b) + (c)"), (23, "all four: the default of mode 1")):
        env = dict(os.environ, ALIVE_DECODER_PRECISION="1", ALIVE_DECODER_BF16_MASK=str(mask))
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": os.path.abspath(ROOT)}], env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:])
            raise SystemExit(f"mask {mask} failed")
        out[str(mask)] = {"groups_on_fp16": what, **json.loads(line[0][7:])}
        print(f"mask {mask:2d} {what:55s} " + "  ".join(f"{k[9:] or 'seed2':5s} {v['rms_error']:.2e}/{v['waveform_rms']:.2f}" for k, v in out[str(mask)].items() if k.startswith("full")), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
