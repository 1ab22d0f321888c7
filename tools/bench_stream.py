"""BASELINE config 5: streaming conversion, 10 ms chunks (-c 160 -b 16 -> 8-frame ring), 50 k-vector library,
per-step latency eager vs hipGraph-captured.  python tools/bench_stream.py [steps] [M] [chunk] [buffersize]"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module.content_encoder import ContentEncoder
from module.decoder import Decoder
from module.f0_estimator import F0Estimator
from module.realtime import RealtimeConverter
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 160
bs = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(7)
tokens = torch.randn(1, 768, M, device=dev, generator=g)
res = {}
for mode in ("eager", "graph"):
    rt = RealtimeConverter(ContentEncoder(seed=2), F0Estimator(seed=2), Decoder(seed=2), tokens, dev, chunk=chunk, buffersize=bs)
    if mode == "graph":
        rt.enable_graph()
    rng = np.random.default_rng(0)
    pcm = (rng.standard_normal(chunk * (steps + bs + 60)) * 3000).astype(np.int16)
    lat = []
    for s in range(steps + bs + 50):
        t0 = time.perf_counter()
        out = rt.step(pcm[s * chunk:(s + 1) * chunk])       # includes H2D of the ring and D2H of the result
        dt = time.perf_counter() - t0
        if out is not None and s >= bs + 50:
            lat.append(dt * 1e3)
    lat = np.array(lat)
    res[mode] = {"p50_ms": round(float(np.percentile(lat, 50)), 3), "p99_ms": round(float(np.percentile(lat, 99)), 3),
                 "mean_ms": round(float(lat.mean()), 3), "rtf": round(float(np.percentile(lat, 50)) / (chunk / 16.0), 4)}
print(json.dumps({"config": f"chunk {chunk} samples ({chunk/16:.0f} ms) x buffersize {bs} = {chunk*bs//320} frames, {M}-vector library, {steps} steps",
                  **res}))
