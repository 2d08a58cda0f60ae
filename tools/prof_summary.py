"""Per-kernel summary of a rocprofv3 --kernel-trace --stats run: python tools/prof_summary.py <dir> [steps]"""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{r['Name'][:88]:88s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / steps / 1e6:9.2f} ms/step "
          f"{float(r['AverageNs']) / 1e3:9.1f} us {float(r['TotalDurationNs']) / tot * 100:5.1f}%")
print(f"total {tot / steps / 1e6:.2f} ms/step")
