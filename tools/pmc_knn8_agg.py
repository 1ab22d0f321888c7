"""aggregate tools/pmc_knn8.sh passes -> profiles/knn_score6_pmc.json: python tools/pmc_knn8_agg.py gpurun_out/pmc8_<tag> [kernel]
(kernel: knn_score6_kernel, the default stage since round 5, or knn_score8_kernel with ALIVE_KNN_PREFILTER=fp8 in the passes)
bench_knn.py launches the scoring kernel 1 (warm) + reps times; counters are averaged per launch."""
import csv, glob, json, os, sys
root = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "knn_score6_kernel"
cnt, ms = {}, []
for f in sorted(glob.glob(os.path.join(root, "pass*", "run_counter_collection.csv"))):
    seen = {}
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"]:
            continue
        seen.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        seen[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE",):
            d_ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            if d_ms > 20.0:                                   # the batch launch, not the probe
                ms.append(d_ms)
    for name, per in seen.items():
        # a search launches the kernel twice since round 2 (the 1 024-frame probe, then the batch): keep the batch launches
        big = [v for v in per.values() if v >= 0.5 * max(per.values())]
        cnt[name] = sum(big) / len(big)
M, Tt = 1_000_000, 172800
fetch_kb, write_kb = cnt.get("FETCH_SIZE", 0.0), cnt.get("WRITE_SIZE", 0.0)
import subprocess
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
except Exception:
    commit = ""
out = {
    "kernel": KERNEL,
    "source_commit": commit,
    "workload": "one launch: 172800 frames (384 windows x 450) x 1,000,000-vector library (tools/bench_knn.py 384 450 1000000 1 biased; tools/pmc_knn8.sh, one counter group per pass); fp6: 450 blocks of 384 frames x 5 library splits, fp8: 675 blocks of 256 frames x 3 splits",
    "launch_ms_profiled": round(sum(ms) / max(1, len(ms)), 1),
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
    "hbm_bytes_formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); separate --pmc passes; Infinity-Cache hits are included in FETCH_SIZE",
    "algorithmic_bytes_per_launch": M * 768 + Tt * 768,
    "algorithmic_bytes_formula": "M*768 (fp8 library once; fp6: the same 32-byte slots per 32 features, 24 bytes of them codes) + T*768 (frames once)",
    "counters_per_launch": {k: v for k, v in sorted(cnt.items())},
}
if "SQ_VALU_MFMA_BUSY_CYCLES" in cnt and "GRBM_GUI_ACTIVE" in cnt:
    out["mfma_pipe_utilisation"] = round(cnt["SQ_VALU_MFMA_BUSY_CYCLES"] / (cnt["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
    out["mfma_pipe_formula"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)"
if cnt.get("SQ_WAVE_CYCLES"):
    for key, name in (("SQ_ACTIVE_INST_VALU", "valu_issue_share_of_wave_cycles"), ("SQ_ACTIVE_INST_LDS", "lds_issue_share_of_wave_cycles"),
                      ("SQ_WAIT_INST_ANY", "wave_cycles_waiting_for_an_instruction")):
        if key in cnt:
            out[name] = round(cnt[key] / cnt["SQ_WAVE_CYCLES"], 3)
if cnt.get("SQ_VALU_MFMA_BUSY_CYCLES") and "SQ_VALU_MFMA_COEXEC_CYCLES" in cnt:
    out["mfma_cycles_with_valu_coexecuting"] = round(cnt["SQ_VALU_MFMA_COEXEC_CYCLES"] / cnt["SQ_VALU_MFMA_BUSY_CYCLES"], 3)
if "TCC_HIT_sum" in cnt:
    out["l2_hit_rate"] = round(cnt["TCC_HIT_sum"] / (cnt["TCC_HIT_sum"] + cnt["TCC_MISS_sum"]), 3)
if "SQ_LDS_BANK_CONFLICT" in cnt:
    out["lds_bank_conflict_share"] = round(cnt["SQ_LDS_BANK_CONFLICT"] / max(1.0, cnt["SQ_LDS_IDX_ACTIVE"]), 4)
print(json.dumps(out, indent=1))
