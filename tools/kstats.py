"""print a rocprofv3 kernel_stats.csv as ms per bench step"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f'{float(r["TotalDurationNs"]) / steps / 1e6:8.2f} ms/step {float(r["Percentage"]):6.2f}% calls {r["Calls"]:>5}  {r["Name"][:120]}')
print("total per step", tot / steps / 1e6)
