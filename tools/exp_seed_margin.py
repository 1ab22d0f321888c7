"""fp8 search with seeded admission (knn.hip, SEED_MARGIN8): kernel time and tier counters per margin.
ALIVE_KNN_SEED_MARGIN=<cosine> python tools/exp_seed_margin.py [kind ...]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module.common import PackedLibrary
dev = "cuda"; N, T, M = 384, 450, 1_000_000
g = torch.Generator(device=dev).manual_seed(1)
lib = PackedLibrary(torch.randn(768, M, device=dev, generator=g))
mode = os.environ.get("EXP_MODE", "fp6")                 # fp6 | fp8 | bf16 | strict
if mode in ("bf16", "fp8", "fp6"): lib = lib.with_prefilter(mode)
if mode == "strict": lib = lib.with_strict()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); b.record()          # (created on first record: the C side records them again)
for kind in (sys.argv[1:] or ["biased", "randn"]):
    src = torch.randn(N, 768, T, device=dev, generator=g)
    if kind == "biased": src = src * 0.2 + torch.randn(1, 768, 1, device=dev, generator=g)
    lib.search(src, 4)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(); lib.search(src, 4, events=(a, b)); t1.record(); torch.cuda.synchronize()
    st = lib.search_stats()
    print(f"{mode} margin {os.environ.get('ALIVE_KNN_SEED_MARGIN', 'default')} {kind}: kernel {a.elapsed_time(b):.2f} ms  search {t0.elapsed_time(t1):.2f} ms  "
          f"seeded blocks {st.get('fp8_blocks_seeded')} / {st.get('bf16_blocks_seeded')}  -> bf16 {st.get('frames_researched_on_bf16')}  fail bf16 {st.get('frames_failed_bf16_certificate')}")
