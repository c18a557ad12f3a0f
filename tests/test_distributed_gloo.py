"""Section 8(e): the N>1 path on CPU - world_size 2, gloo backend, 127.0.0.1 rendezvous.
Each rank inspects image_paths[r::2] with a mock backend, one all_gather exchanges the records,
every rank returns the full input-ordered result."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, paths, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vision_inspection_system_amd import config as C
    from vision_inspection_system_amd.batch import gather_records, run_multi_image_inspection
    C.set_config(C.Config(vlm_inspector_provider="mock", vlm_auditor_provider="mock"))
    # raw collective: ragged payloads, one rank empty
    recs = gather_records([{"r": rank, "i": i, "pad": "x" * (50 * rank)} for i in range(3 * rank)], world)
    assert [(r["r"], r["i"]) for r in recs] == [(1, 0), (1, 1), (1, 2)]

    def fake_inspect(image_path, criticality, domain, user_notes):
        if image_path.endswith("bad.png"):
            raise RuntimeError("decode error")
        return {"inspector_result": {"who": rank}, "auditor_result": {}, "safety_verdict": {"verdict": "SAFE"},
                "consensus": {"combined_defects": [{"safety_impact": "COSMETIC"}]}, "processing_time": 0.01}
    out = run_multi_image_inspection(paths, session_id=None, _inspect=fake_inspect)
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f, default=str)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_batch(tmp_path):
    paths = [str(tmp_path / n) for n in ("a.png", "b.png", "bad.png", "c.png", "d.png")]
    port = _free_port()
    mp.spawn(_worker, args=(2, port, paths, str(tmp_path)), nprocs=2, join=True)
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    assert r0["image_results"] == r1["image_results"] and r0["session_id"] == r1["session_id"]
    res = r0["image_results"]
    assert [v["image_path"] for v in res.values()] == paths          # input order restored
    assert [v.get("completed") for v in res.values()] == [True, True, False, True, True]
    # rank r handled paths[r::2]
    assert [v["inspector_result"]["who"] for v in res.values() if v["completed"]] == [0, 1, 1, 0]
    sr = r0["session_results"]
    assert sr["total_images"] == 5 and sr["completed_images"] == 4 and sr["failed_images"] == 1
    assert sr["aggregate_verdict"] == "SAFE" and sr["cosmetic_defects"] == 4
    assert res[list(res)[2]]["error"] == "decode error"
