"""Register / LDS / spill metadata of every kernel of the product library, from the gfx950 listings of csrc/*.hip with the Makefile's flags:
python tools/kernel_meta.py [out.json]   (CPU only; the numbers DESIGN.md quotes per kernel come from here)"""
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "alive-vc_amd", "csrc")


def main():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    srcs = [ln for ln in mk.splitlines() if ln.startswith("SRCS")][0].split(":=")[1].split()
    mid = [ln for ln in mk.splitlines() if ln.startswith("FLAGS_filter_mid")][0].split(":=")[1].split()
    base = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function", "--cuda-device-only", "-S"]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        def build(src):
            o = os.path.join(tmp, src + ".s")
            extra = mid if src in ("filter_mid.hip", "filter_small.hip") else []
            subprocess.run(["/opt/rocm/bin/hipcc"] + base + extra + [os.path.join(CSRC, src), "-o", o], check=True, capture_output=True, timeout=1800)
            return src, o
        with ThreadPoolExecutor(6) as ex:
            for src, path in ex.map(build, srcs):
                t = open(path).read()
                if "amdhsa.kernels" not in t:
                    continue
                meta = t[t.index("amdhsa.kernels"):]
                for blk in meta.split("  - .agpr_count:")[1:]:
                    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                    dem = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", dem))
                    dem = re.sub(r"\(.*$", "", dem)
                    g = lambda k: int((re.search(k + r":\s+(\d+)", blk) or [None, "0"])[1])
                    vg, lds = g(r"\.vgpr_count"), g(r"\.group_segment_fixed_size")
                    alloc = (vg + 7) // 8 * 8
                    out[dem] = {"file": src, "registers_vgpr_plus_agpr": vg, "agpr": int(blk.split("\n")[0].strip()), "sgpr": g(r"\.sgpr_count"),
                                "static_lds_bytes": lds, "spilled_vgprs": g(r"\.vgpr_spill_count"), "scratch_bytes": g(r"\.private_segment_fixed_size"),
                                "waves_per_simd_by_registers": min(8, 512 // max(alloc, 1))}
    for k in sorted(out):
        v = out[k]
        print(f"{k[:78]:78s} {v['file']:18s} regs {v['registers_vgpr_plus_agpr']:3d} (agpr {v['agpr']:3d}) sgpr {v['sgpr']:3d} lds {v['static_lds_bytes']:6d} spill {v['spilled_vgprs']} waves/SIMD {v['waves_per_simd_by_registers']}")
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
