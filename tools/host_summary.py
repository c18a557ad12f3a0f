"""One line per JSON line of a `bench.py --workload batch256 --dry-device cpu --dry-ingest` log: ranks, images/s, threads, per-rank wall."""
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print(d["n_gpus"], "ranks:", round(d["value"], 1), "images/s  ", d["config"]["images"], "images, ingest threads/rank",
              d["config"]["ingest_threads_per_rank"], "host_cpus", d["config"]["host_cpus"], " per-rank call_s",
              [round(p["call_s"], 2) for p in d["per_rank"]])
    elif ln.strip():
        print("nproc", ln.strip())
