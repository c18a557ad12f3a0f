"""bench.py --gpus N must start its own ranks when nothing launched them: the parent spawns ONE
torch.distributed.run child before touching the GPU and relays rank 0's JSON line.  Rehearsed on CPU with the gloo
backend and no model (--dry-device cpu): launcher, 127.0.0.1 rendezvous, record gather, barriers, max over ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--model", "tiny",
                           "--dry-device", "cpu", "--steps", "3", "--warmup", "1"] + extra,
                          capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    r = _run(["--gpus", "2", "--batch", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dry"] is True
    assert out["config"]["parallelism"] == "dp2" and out["value"] > 0


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_launcher_propagates_failure():
    # an impossible option makes every rank exit non-zero: the parent must report failure, not print a result
    r = _run(["--gpus", "2", "--batch", "not-a-number"])
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_the_product_path_refuses_gloo_and_cpu():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--model", "tiny"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "dry-device" in (r.stderr + r.stdout)


def test_batch256_workload_three_ranks_on_cpu():
    """`bench.py --gpus 3 --workload batch256` (BASELINE configs[3] as a workload), rehearsed with canned-response agents:
    rank 0 writes the PNG files, the directory is agreed through the store, every rank calls run_batch_inspection on the
    full list, rank r inspects paths[r::3], one gather, strong-scaling JSON line with a per-rank breakdown."""
    r = _run(["--gpus", "3", "--workload", "batch256", "--images", "14"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 3 and out["scaling"] == "strong" and out["dry"] is True and out["value"] > 0
    assert out["config"]["images"] == 14 and out["config"]["completed"] == 14
    assert [p["images"] for p in out["per_rank"]] == [5, 5, 4] and [p["rank"] for p in out["per_rank"]] == [0, 1, 2]


def test_batch256_host_only_ingest_two_ranks_on_cpu():
    """`--dry-ingest`: the Inspector on the LOCAL provider with an engine stand-in - the product's ingest path (a3 encode on
    the pool, base64 + Huffman decode of every data URI, tokenisation, parse, consensus, gates) with no model and no GPU: the
    host ceiling of a rank.  Two gloo ranks, every image completed, nothing counted as engine time."""
    r = _run(["--gpus", "2", "--workload", "batch256", "--images", "10", "--dry-ingest", "--model", "7b", "--image-size", "256"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["dry"] is True and out["n_gpus"] == 2 and out["config"]["completed"] == 10
    assert "host-only stand-in" in out["config"]["workload"]
    assert [p["images"] for p in out["per_rank"]] == [5, 5]
    assert all(p["engine_device_s"] == 0.0 for p in out["per_rank"])
