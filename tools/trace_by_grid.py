"""Median / min duration per (kernel, workgroups) from a rocprofv3 --kernel-trace directory, optionally only kernels whose
name contains one of the given substrings:  python tools/trace_by_grid.py DIR [substr ...] [--cycle SUBSTR:l1,l2,...]
--cycle: launches of kernels matching SUBSTR are labelled l1, l2, ... cyclically in launch order (a decode step's projections
repeat qkv, o, gate/up, down per layer and end with the lm_head: two of them share a grid size)."""
import collections, csv, glob, sys
args = sys.argv[2:]
cycle_sub, cycle_labels = None, []
if "--cycle" in args:
    i = args.index("--cycle")
    cycle_sub, lab = args[i + 1].split(":")
    cycle_labels = lab.split(",")
    del args[i:i + 2]
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
want = args
d = collections.defaultdict(list)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
k = 0
for r in rows:
    n = r["Kernel_Name"]
    if want and not any(w in n for w in want):
        continue
    wg = max(1, int(r.get("Workgroup_Size_X", 1)))
    name = n.split("(")[0].replace("void ", "")[-48:]
    if cycle_sub and cycle_sub in n:
        name = (name + " " + cycle_labels[k % len(cycle_labels)])[-48:]
        k += 1
    d[(name, int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]))].append(
        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in d.values())
print(f"{'kernel':50s} {'wg_x':>6s} {'gy':>4s} {'calls':>7s} {'median_us':>10s} {'min_us':>8s} {'total_ms':>9s} {'share':>6s}")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k[0]:50s} {k[1]:6d} {k[2]:4d} {len(v):7d} {v[len(v) // 2] / 1e3:10.1f} {v[0] / 1e3:8.1f} {sum(v) / 1e6:9.3f} {sum(v) / tot:6.3f}")
