"""Sum rocprofv3 --pmc counters per kernel name: python tools/pmc_agg.py <dir> [name filter]."""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if flt in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(calls[k])
    print(k, "launches", n, {a: round(b / n / 1e6, 3) for a, b in sorted(v.items())}, "(millions per launch)")
