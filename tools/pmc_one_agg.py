"""per-kernel sums of the counters collected by tools/pmc_one.sh: python tools/pmc_one_agg.py gpurun_out/pmc1_<tag> [substring]"""
import csv, glob, sys, collections
root = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob(root + "/pass*/run_counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        if want not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (f, r["Dispatch_Id"]) not in seen:
            seen.add((f, r["Dispatch_Id"])); calls[(k, f)] += 1
for k, d in acc.items():
    n = max(v for (kk, f), v in calls.items() if kk == k)
    print(k, "dispatches per pass:", n)
    for c, v in sorted(d.items()): print(f"   {c:32s} {v / n:16.0f} per dispatch")
