"""Per-kernel sums of rocprofv3 --pmc counters: python tools/pmc_summary.py <dir> [<dir> ...]
Prints, per kernel name, calls and each counter's per-call mean (FETCH_SIZE / WRITE_SIZE are reported by
rocprofv3 in KiB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads)."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(set))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(agg, key=lambda k: -max(agg[k].values())):
    parts = [f"{c}: total {v:.4g} per-call {v / max(len(calls[k][c]), 1):.4g} (n={len(calls[k][c])})" for c, v in agg[k].items()]
    print(k, "|", " | ".join(parts))
