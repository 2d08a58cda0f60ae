import csv, glob, collections, sys
f = glob.glob("gpurun_out/trace1/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_bf16_kernel" in r["Kernel_Name"]]
n = len(rows) // 2
rows = rows[n:]   # second step
tot = 0
out = []
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    out.append((d, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"), r["Kernel_Name"][:40]))
print("launches", len(rows), "total us", tot)
for i, o in enumerate(out):
    print(i, "%.1f" % o[0], o[1])
