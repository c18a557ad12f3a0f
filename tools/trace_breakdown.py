"""Per-(kernel, grid) table of a rocprofv3 --kernel-trace CSV, normalised per image: launches per image (may be fractional
for stacked / shared passes), median / min duration, ms per image, share.  Unlike tools/prefill_breakdown.py it keeps every
kernel whose name matches --include (default: all) and is not matched by --exclude, so it serves batched runs (fp8 batch 4,
mllama batch 32, dual) where launches per image are not integers.

  python tools/trace_breakdown.py TRACE_DIR N_IMAGES out.csv [--exclude a,b,c] [--include x,y] [--skip-first K]

--skip-first K drops the first K launches of every (kernel, grid) row (warm-up step) when K launches exist."""
import argparse, collections, csv, glob, sys

ap = argparse.ArgumentParser()
ap.add_argument("trace_dir")
ap.add_argument("n_images", type=float)
ap.add_argument("out", nargs="?")
ap.add_argument("--exclude", default="at::native,__amd_rocclr,Custom_Cijk")
ap.add_argument("--include", default="")
ap.add_argument("--top", type=int, default=24)
a = ap.parse_args()
f = glob.glob(a.trace_dir + "/**/*kernel_trace.csv", recursive=True)[0]
exc = [x for x in a.exclude.split(",") if x]
inc = [x for x in a.include.split(",") if x]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(k in n for k in exc) or (inc and not any(k in n for k in inc)):
        continue
    name = n.split("(")[0].replace("void ", "")
    wg = int(r.get("Workgroup_Size_X", 1) or 1)
    d[(name, int(r["Grid_Size_X"]) // max(wg, 1), int(r["Grid_Size_Y"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for (name, gx, gy), v in d.items():
    v.sort()
    rows.append((sum(v) / a.n_images / 1e3, name, gx, gy, len(v) / a.n_images, v[len(v) // 2], v[0]))
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
out = [["kernel", "workgroups_x", "grid_y", "launches_per_image", "median_us", "min_us", "ms_per_image", "share"]]
for ms, name, gx, gy, n, med, mn in rows:
    out.append([name, gx, gy, f"{n:.2f}", f"{med:.1f}", f"{mn:.1f}", f"{ms:.3f}", f"{ms / total:.4f}"])
out.append(["TOTAL (kernel time, no gaps)", "", "", "", "", "", f"{total:.3f}", "1"])
w = csv.writer(open(a.out, "w", newline="") if a.out else sys.stdout)
w.writerows(out)
if a.out:
    for r in out[:a.top] + out[-1:]:
        print(",".join(str(x) for x in r))
