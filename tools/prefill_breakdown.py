"""Per-image prompt-pass (prefill) breakdown from a rocprofv3 --kernel-trace CSV of `python bench.py`:
one row per (kernel, grid): launches per image, median / min duration, ms per image, share of the prompt pass.
Decode kernels (gemv / decode_attn / argmax / gather) and torch's own fill kernels are left out.

  python tools/prefill_breakdown.py TRACE_DIR N_PREFILLS [out.csv]
"""
import collections, csv, glob, sys

DECODE = ("gemv_", "decode_attn", "argmax_", "gather_rows", "at::native", "__amd_rocclr", "scatter_rows", "gemm_decode", "skinny_",
          "Custom_Cijk")      # + the library GEMM of bench.py's peak microbenchmark
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n_img = int(sys.argv[2])
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(k in n for k in DECODE):
        continue
    name = n.split("(")[0].replace("void ", "")
    wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else 1
    d[(name, int(r["Grid_Size_X"]) // max(wg, 1), int(r["Grid_Size_Y"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for (name, gx, gy), v in d.items():
    if len(v) % n_img:      # not once (or k times) per prompt pass: bench.py's 8192^3 microbenchmark launches
        continue
    v.sort()
    rows.append((sum(v) / n_img / 1e3, name, gx, gy, len(v) / n_img, v[len(v) // 2], v[0]))
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
out = [["kernel", "workgroups_x", "grid_y", "launches_per_image", "median_us", "min_us", "ms_per_image", "share"]]
for ms, name, gx, gy, n, med, mn in rows:
    out.append([name, gx, gy, f"{n:g}", f"{med:.1f}", f"{mn:.1f}", f"{ms:.3f}", f"{ms / total:.4f}"])
out.append(["TOTAL (kernel time, no gaps)", "", "", "", "", "", f"{total:.3f}", "1"])
w = csv.writer(open(sys.argv[3], "w", newline="") if len(sys.argv) > 3 else sys.stdout)
w.writerows(out)
if len(sys.argv) > 3:
    for r in out[:18] + out[-1:]:
        print(",".join(str(x) for x in r))
