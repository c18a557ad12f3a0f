"""Condense rocprofv3 output directories (gpurun_out/...) into the small, committed files under profiles/.

  python tools/summarize_profile.py --round r01 --stats gpurun_out/prof2/stats --fetch gpurun_out/prof2/fetch \
        --write gpurun_out/prof2/write --bench-log gpurun_out/prof2/bench_stats.log

Writes profiles/<round>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary of the default bench command),
profiles/<round>_gemv_by_shape.csv (per-GEMV-shape durations from the trace) and
profiles/<round>_gemv_traffic.json (HBM bytes per gemv launch from the FETCH_SIZE / WRITE_SIZE passes, with the
gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2).
"""
import argparse
import collections
import csv
import glob
import json
import os


def one(pattern):
    hits = glob.glob(pattern, recursive=True)
    if not hits:
        raise SystemExit(f"no file matches {pattern}")
    return hits[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--bench-log")
    ap.add_argument("--out", default="profiles")
    ap.add_argument("--traffic-kernel", default="gemv_bf16_kernel", help="kernel (name prefix) of the FETCH / WRITE passes")
    ap.add_argument("--traffic-name", default="gemv", help="profiles/<round>_<name>_traffic.json")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.stats:
        rows = list(csv.DictReader(open(one(os.path.join(a.stats, "**", "*_kernel_stats.csv")))))
        with open(os.path.join(a.out, f"{a.round}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                if "at::native" in r["Name"] and float(r["Percentage"]) < 0.05:
                    continue
                w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                            r["MaxNs"]])
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(one(os.path.join(a.stats, "**", "*_kernel_trace.csv")))):
            if "gemv_bf16_kernel" in r["Kernel_Name"]:      # <1, false>: projections; <1, true>: lm_head with the pick's first stage
                agg[int(r["Grid_Size_X"]) // 256].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        with open(os.path.join(a.out, f"{a.round}_gemv_by_shape.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["workgroups", "launches", "avg_us", "median_us", "min_us"])
            for k, v in sorted(agg.items()):
                v2 = sorted(v)
                w.writerow([k, len(v), f"{sum(v) / len(v) / 1e3:.2f}", f"{v2[len(v2) // 2] / 1e3:.2f}", f"{v2[0] / 1e3:.2f}"])
    if a.fetch and a.write:
        kernels = [k for k in a.traffic_kernel.split(",") if k]     # several substrings: the launches of all of them together
        by_kernel = {}

        def per_launch(d, name):
            vals = []
            for r in csv.DictReader(open(one(os.path.join(d, "**", "*_counter_collection.csv")))):
                hit = next((k for k in kernels if k in r["Kernel_Name"]), None)
                if r["Counter_Name"] == name and hit:
                    vals.append(float(r["Counter_Value"]))
                    e = by_kernel.setdefault(hit, {}).setdefault(name, [0.0, 0])
                    e[0] += float(r["Counter_Value"]); e[1] += 1
            return sum(vals) / len(vals), len(vals)
        fetch_kb, n1 = per_launch(a.fetch, "FETCH_SIZE")
        write_kb, n2 = per_launch(a.write, "WRITE_SIZE")
        out = {"kernel": a.traffic_kernel, "launches_sampled": [n1, n2], "FETCH_SIZE_KB_per_launch": fetch_kb,
               "WRITE_SIZE_KB_per_launch": write_kb,
               "by_kernel_KB_per_launch": {k: {c: v[0] / v[1] for c, v in d.items()} | {"launches": max(v[1] for v in d.values())}
                                           for k, d in by_kernel.items()},
               "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads -> x2 "
                             "(MI355X_MICROARCH.md, HBM); WRITE_SIZE taken as is",
               "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0}
        with open(os.path.join(a.out, f"{a.round}_{a.traffic_name}_traffic.json"), "w") as f:
            json.dump(out, f, indent=1)
    if a.bench_log:
        for line in open(a.bench_log):
            if line.startswith("{"):
                with open(os.path.join(a.out, f"{a.round}_bench_under_rocprof.json"), "w") as f:
                    f.write(line)


if __name__ == "__main__":
    main()
