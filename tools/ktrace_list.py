"""per-dispatch listing of a rocprofv3 kernel trace (csv): python tools/ktrace_list.py <*_kernel_trace.csv> [skip_first_n]
prints dispatch order, duration, grid and a short kernel name -- which conv of the decoder costs what"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tot = 0.0
for i, r in enumerate(rows[skip:]):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    grid = "x".join(str(int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"]))) for a in "XYZ")
    print(f"{i:4d} {us:9.1f} us  grid {grid:>16s}  {name}")
print(f"total {tot / 1e3:.2f} ms")
