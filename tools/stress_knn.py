import os, sys, random, torch
sys.path.insert(0, os.path.join(os.getcwd(), "alive-vc_amd"))
from module.common import PackedLibrary
dev = "cuda"; random.seed(7); bad = 0
g = torch.Generator(device=dev).manual_seed(99)
for it in range(40):
    m = random.choice([9, 40, 300, 1023, 1025, 5000, 33333, 120000, 400000])
    n = random.choice([1, 2, 5, 17]); t = random.choice([1, 3, 64, 255, 450, 1000])
    k = random.choice([1, 2, 4, 8]); k = min(k, m)
    kind = random.choice(["randn", "clustered", "dups"])
    lib = torch.randn(768, m, device=dev, generator=g)
    src = torch.randn(n, 768, t, device=dev, generator=g)
    if kind == "clustered":
        base = torch.randn(768, 1, device=dev, generator=g)
        lib = base + 0.3 * lib; src = base.unsqueeze(0) + 0.3 * src
    if kind == "dups" and m > 20:
        lib[:, m // 2:m // 2 + 10] = lib[:, :10]
    l8, l16 = PackedLibrary(lib, prefilter="fp8"), PackedLibrary(lib, prefilter="bf16")
    v8, i8 = l8.search(src, k); v16, i16 = l16.search(src, k)
    ok = torch.equal(v8, v16) and (torch.equal(i8, i16) or kind == "dups")
    if kind == "dups" and not torch.equal(i8, i16):
        ok = torch.equal(v8, v16)     # tied duplicates may swap
    print(it, m, n, t, k, kind, "ok" if ok else "MISMATCH", "researched", l8.fallback_frames())
    bad += 0 if ok else 1
print("mismatching configs:", bad)
