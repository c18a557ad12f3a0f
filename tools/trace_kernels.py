"""Median/min duration per (kernel, grid) from a rocprofv3 --kernel-trace CSV: python tools/trace_kernels.py DIR [substr ...]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
keys = sys.argv[2:]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if not keys or any(k in n for k in keys):
        d[(n.split("(")[0][-44:], r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k[0]:44s} grid=({k[1]},{k[2]},{k[3]}) n={len(v):5d} med {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}  total {sum(v)/1e3:8.2f} ms")
