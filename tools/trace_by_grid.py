"""Median / min duration per (kernel, workgroups) from a rocprofv3 --kernel-trace directory, optionally only kernels whose
name contains one of the given substrings:  python tools/trace_by_grid.py DIR [substr ...]"""
import collections, csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
want = sys.argv[2:]
d = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"]
    if want and not any(w in n for w in want):
        continue
    wg = max(1, int(r.get("Workgroup_Size_X", 1)))
    d[(n.split("(")[0].replace("void ", "")[-48:], int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]))].append(
        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in d.values())
print(f"{'kernel':50s} {'wg_x':>6s} {'gy':>4s} {'calls':>7s} {'median_us':>10s} {'min_us':>8s} {'total_ms':>9s} {'share':>6s}")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k[0]:50s} {k[1]:6d} {k[2]:4d} {len(v):7d} {v[len(v) // 2] / 1e3:10.1f} {v[0] / 1e3:8.1f} {sum(v) / 1e6:9.3f} {sum(v) / tot:6.3f}")
