"""Experiment (round 3): the step as `groups` independent pipelines (spectrogram -> encoders -> kNN match -> decoder over a
slice of the windows), each on a stream of its own, so that one group's memory- / VALU-bound kernels can run beside another
group's MFMA-bound scoring kernel (360 VGPRs and 118 KB of LDS per block: kernels with <= 152 VGPRs and <= 42 KB fit on the
same CU).  python tools/exp_pipelines.py [groups ...]"""
import os, sys, time, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "alive-vc_amd")); sys.path.insert(0, ROOT)
os.environ["ALIVE_STREAMS"] = os.environ.get("ALIVE_STREAMS", "1")
import bench
from module.common import PackedLibrary
from module.content_encoder import ContentEncoder
from module.decoder import Decoder
from module.f0_estimator import F0Estimator
from module.pipeline import Converter
dev = torch.device("cuda")
M = 1_000_000
tokens = torch.randn(768, M, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
conv = Converter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), dev).set_library(PackedLibrary(tokens))
windows = bench.synth_windows(64, 10.0, 48000, dev, seed=100)
n = windows.shape[0]
pool = [torch.cuda.Stream(device=dev) for _ in range(8)]
def step(groups, wb):
    if groups == 1:
        return conv.convert_windows(windows, k=4, window_batch=wb)
    out = torch.empty_like(windows)
    cur = torch.cuda.current_stream()
    per = (n + groups - 1) // groups
    for g in range(groups):
        st = pool[g]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            out[g * per:(g + 1) * per] = conv.convert_windows(windows[g * per:(g + 1) * per], k=4, window_batch=wb)
    for g in range(groups):
        cur.wait_stream(pool[g])
    return out
ref = step(1, 128); torch.cuda.synchronize()
for groups in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6]:
    for wb in (128, 64):
        o = step(groups, wb); torch.cuda.synchronize()
        same = torch.equal(o, ref)
        t0 = time.perf_counter()
        for _ in range(4): step(groups, wb)
        torch.cuda.synchronize()
        print(f"groups {groups} window_batch {wb}: {(time.perf_counter() - t0) / 4 * 1e3:.1f} ms  bitwise equal to one pipeline: {same}", flush=True)
