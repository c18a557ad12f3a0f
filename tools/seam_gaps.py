"""GPU-idle time inside the prompt-pass phase of a run_batch_inspection call, from a rocprofv3
--kernel-trace of `python3 tools/ingest_bench.py --images N --auditor mock --threads T`:

  python tools/seam_gaps.py TRACE_DIR N [OUT.json]

The measured call is the LAST one of the trace: its prompt-pass phase runs from the first patchify launch of its N images
(one per image) to the first launch of its shared decode loop (the first batched decode attention after them).  Prints the
phase's wall time, the time some kernel was running (union of kernel intervals: prompt passes of a ViT group alternate
between two streams), the idle time, and the idle time broken down by the size of the gap - sub-20 us gaps are kernel
boundaries, the long ones are the launch thread not having the next kernel queued (host-bound) or waiting for a request's
decode."""
import csv, glob, json, sys

d, n_img = sys.argv[1], int(sys.argv[2])
out_path = sys.argv[3] if len(sys.argv) > 3 else None
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_Y"])) for r in csv.DictReader(open(f))]
rows.sort()
pat = [i for i, r in enumerate(rows) if "patchify_u8" in r[2]]
assert len(pat) >= n_img, f"only {len(pat)} patchify launches in the trace"
i0 = pat[-n_img]
i1 = next(i for i in range(pat[-1], len(rows)) if "decode_attn" in rows[i][2] and rows[i][3] > 1)
phase = rows[i0:i1]
t0, t1 = phase[0][0], rows[i1][0]
busy, cur_s, cur_e = 0, phase[0][0], phase[0][1]
gaps = []
for s, e, _, _ in phase[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
if t1 > cur_e:
    gaps.append(t1 - cur_e)
bins = {"< 20 us (kernel boundaries)": (0, 20e3), "20 us - 1 ms": (20e3, 1e6), "1 - 10 ms": (1e6, 1e7), ">= 10 ms": (1e7, 1e18)}
res = {"images": n_img, "kernels": len(phase), "phase_ms": (t1 - t0) / 1e6, "gpu_busy_ms": busy / 1e6, "gpu_idle_ms": sum(gaps) / 1e6,
       "idle_by_gap_size": {k: {"gaps": sum(1 for g in gaps if lo <= g < hi), "ms": sum(g for g in gaps if lo <= g < hi) / 1e6}
                            for k, (lo, hi) in bins.items()},
       "sum_of_kernel_durations_ms": sum(e - s for s, e, _, _ in phase) / 1e6,
       "largest_gaps_ms": [g / 1e6 for g in sorted(gaps, reverse=True)[:8]]}
print(json.dumps(res, indent=1))
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
