"""experiment: three decoder batches of 128 windows one after the other on one stream vs on three streams (own workspaces)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module.decoder import Decoder
dev = "cuda"
decs = [Decoder(seed=2).to(dev) for _ in range(3)]
g = torch.Generator(device=dev).manual_seed(5)
feats = [torch.randn(128, 768, 450, device=dev, generator=g) for _ in range(3)]
f0s = [100.0 + 50.0 * torch.rand(128, 1, 450, device=dev, generator=g) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]


def seq():
    return [decs[0](feats[i], f0s[i])[0] for i in range(3)]


def par():
    outs = [None] * 3
    cur = torch.cuda.current_stream()
    for i in range(3):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            outs[i] = decs[i](feats[i], f0s[i])[0]
    for i in range(3):
        cur.wait_stream(streams[i])
    return outs


for name, fn in (("sequential", seq), ("3 streams", par), ("sequential", seq), ("3 streams", par)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        o = fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per 384 windows")
a, b = seq(), par()
torch.cuda.synchronize()
print("same results:", all(torch.equal(x, y) for x, y in zip(a, b)))
