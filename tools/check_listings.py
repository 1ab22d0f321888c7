"""Build-time guard (csrc/Makefile runs it as part of `make all`): two properties the results depend on are enforced by
nothing but the shape of the code hipcc emits, so the BUILD -- not only pytest -- compiles the four guarded sources to gfx950 listings
and fails when either is violated:
  * every MFMA accumulation chain that starts beside a finished, still unread accumulator keeps >= 8 idle wait states to it
    (common.h ALIVE_CHAIN_GAP; DESIGN.md 3.2b': fused FilterBlocks, block-scaled scoring kernels);
  * nothing touches the registers of the fp6 scoring kernel's asm `ds_read_b64` fragment halves before an lgkmcnt wait
    (knn.hip: hipcc does not know those reads exist).
usage: python tools/check_listings.py [csrc dir]      exit status 0 = clean"""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import mfma_hazard_scan as hz  # noqa: E402


def main():
    csrc = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "..", "alive-vc_amd", "csrc")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
            "--cuda-device-only", "-S"]
    mk = open(os.path.join(csrc, "Makefile")).read()
    mid = [ln for ln in mk.splitlines() if ln.startswith("FLAGS_filter_mid")][0].split(":=")[1].split()
    jobs = {"knn": ("knn.hip", []), "filter_mid": ("filter_mid.hip", mid), "filter_small": ("filter_small.hip", mid),
            "filter_big": ("filter_big.hip", [])}
    least = {"filter_big": 32}                      # MFMAs a listing must show at least (else the scan looked at the wrong thing): 100 elsewhere
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        def build(item):
            tag, (src, extra) = item
            out = os.path.join(tmp, tag + ".s")
            subprocess.run([hipcc] + base + extra + [os.path.join(csrc, src), "-o", out], check=True, capture_output=True, timeout=1800)
            return tag, out
        with ThreadPoolExecutor(4) as ex:
            lst = dict(ex.map(build, jobs.items()))
        for tag, path in lst.items():
            r = hz.chain_gap_scan(path)
            short = [s for s in r["switches"] if s[4] < hz.CHAIN_GAP_MIN]
            print(f"check_listings: {tag}: {r['mfma']} MFMAs, {len(r['switches'])} chain switches beside an unread accumulator, {len(short)} without the gap")
            if r["mfma"] < least.get(tag, 100) or short:
                bad.append((tag, "chain gap", short[:3]))
        for kern in ("knn_score6_kernel", "knn_probe6_kernel"):
            n_asm, unwaited = hz.asm_lds_reads_are_waited_for(lst["knn"], kern)
            print(f"check_listings: {kern}: {n_asm} asm LDS reads, {unwaited} touched before an lgkmcnt wait")
            if n_asm != 36 or unwaited:
                bad.append((kern, "asm ds_read", n_asm, unwaited))
    if bad:
        print("check_listings: FAILED", bad, file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
