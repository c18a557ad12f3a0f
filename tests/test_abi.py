"""The C-ABI library loads on a GPU-less host and exports every symbol include/vis_hip.h declares.
No compute calls here (no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    p = os.path.join(ROOT, "vision-inspection-system_amd", "csrc", "libvis_hip.so")
    if not os.path.exists(p):
        import __graft_entry__ as g
        g.build()
    return p


def test_header_symbols_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    header = open(os.path.join(ROOT, "include", "vis_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(vis_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 27
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in vis_hip.h but not exported"
    lib.vis_abi_version.restype = ctypes.c_int
    assert lib.vis_abi_version() == 3


def test_binding_covers_header(lib_path):
    from vision_inspection_system_amd import hip
    header = open(os.path.join(ROOT, "include", "vis_hip.h")).read()
    declared = set(re.findall(r"\b(vis_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hip.exported_symbols())
    hip.load()


def test_binding_signatures_match_header(lib_path):
    """Every ctypes signature of hip.py against the prototype in include/vis_hip.h, argument by argument (pointer / int /
    float / unsigned / long long).  A binding one int short is a ctypes.ArgumentError at the first call - on the GPU box, after
    minutes of queueing (r05) - or, worse, shifted arguments; this is the CPU-side check."""
    from vision_inspection_system_amd import hip
    header = open(os.path.join(ROOT, "include", "vis_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = dict((m.group(1), m.group(2)) for m in re.finditer(r"\b(vis_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", header))

    def code(param: str) -> str:
        param = " ".join(param.split())
        if "*" in param or param.startswith("vis_stream_t") or param.startswith("hipStream_t"):
            return "p"
        for prefix, c in (("long long", "l"), ("unsigned", "u"), ("float", "f"), ("int", "i")):
            if param.startswith(prefix + " ") or param == prefix:
                return c
        raise AssertionError(f"unparsed parameter {param!r}")

    for name, sig in hip._SIGS.items():
        assert name in protos, name
        params = [p for p in protos[name].split(",") if p.strip() and p.strip() != "void"]
        want = "".join(code(p) for p in params)
        assert sig == want, f"{name}: hip.py binds {sig!r} ({len(sig)} arguments), vis_hip.h declares {want!r} ({len(want)})"


def test_launchers_reject_bad_arguments_without_gpu(lib_path):
    """Argument validation happens before any HIP call, so it is testable without a device."""
    from vision_inspection_system_amd import hip
    lib = hip.load()
    assert lib.vis_gemm_bf16(None, None, None, None, None, 1, 1, 64, 64, 64, 4, 0, 0, None) == 1
    assert lib.vis_gemm_bf16(16, 16, None, None, 16, 4, 4, 40, 40, 40, 4, 0, 0, None) == 1      # K % 64
    assert lib.vis_rmsnorm_bf16(16, 16, 16, 1, 6000, 6000, 6000, 1e-6, None) == 1                 # row too long
    assert lib.vis_attn_prefill(16, 16, 16, 16, 16, 1, 4, 3, 128, 8, 8, 64, 512, 1, 0.1, None) == 1  # Hq % Hkv
    assert lib.vis_qkv_rope_split(16, 16, 16, 16, 16, None, None, 8, 1536, 4, 4, 96, 8, 0, 0, None) == 1  # head dim


def test_no_cpu_fallback():
    """The product path refuses to run without a GPU instead of silently computing elsewhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vision_inspection_system_amd import hip
    from vision_inspection_system_amd.config import Qwen2VLConfig
    from vision_inspection_system_amd.engine import Qwen2VLEngine
    with pytest.raises(hip.HipLibraryError):
        Qwen2VLEngine(Qwen2VLConfig.tiny(), None, "cuda:0")
    with pytest.raises(hip.HipLibraryError):
        hip.gemm(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "vision-inspection-system_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{fn} imports the oracle"
            assert "qwen2vl_ref" not in src, f"{fn} references the oracle module"


def test_decode_gemm_slot_geometry_without_gpu(lib_path):
    """vis_gemm_decode_ksplit is host-only arithmetic: the partial-slot count the stream-K cut needs."""
    from vision_inspection_system_amd import hip
    lib = hip.load()
    for N, K in [(4608, 3584), (3584, 3584), (37888, 3584), (3584, 18944), (152064, 3584), (512, 256), (1000, 704)]:
        slots = lib.vis_gemm_decode_ksplit(N, K)
        assert 1 <= slots <= 16, (N, K, slots)
    assert lib.vis_gemm_decode_ksplit(0, 64) == 0 and lib.vis_gemm_decode_ksplit(128, 32) == 0
    # fewer slots than the geometry needs is an argument error (checked before any HIP call)
    need = lib.vis_gemm_decode_ksplit(3584, 18944)
    assert need > 1
    assert lib.vis_gemm_decode_bf16(16, 16, 16, None, 8, 3584, 18944, 18944, 18944, 0, need - 1, 0, None) == 1


def test_attention_work_planner_partitions_rows():
    """plan_attn_items (host): every query row of every segment is covered exactly once, items never straddle a
    segment, 128-row items come first, and the ViT single-image case trades trailing full items for halves."""
    from vision_inspection_system_amd import hip
    cases = [([(0, 4900)], 16), ([(0, 100), (100, 356)], 4), ([(i * 64, (i + 1) * 64) for i in range(5)], 4),
             ([(0, 4900), (4900, 9800), (9800, 11000)], 16), ([(0, 37)], 16), ([(0, 4900)], 0)]
    for segs, heads in cases:
        items = hip.plan_attn_items(segs, heads)
        seen = {}
        for (q0, qn, k0, k1) in items:
            assert 0 < qn <= 128 and (k0, k1) in segs and k0 <= q0 and q0 + qn <= k1
            for r in range(q0, q0 + qn):
                assert r not in seen
                seen[r] = 1
        assert len(seen) == sum(e - s for s, e in segs)
        sizes = [it[1] > 64 for it in items]
        assert sizes == sorted(sizes, reverse=True)          # full items first
    vit = hip.plan_attn_items([(0, 4900)], 16)
    assert hip.ATTN_SLOTS - 32 < len(vit) * 16 <= hip.ATTN_SLOTS - 16      # as fine as one round allows, one item of slack
    assert hip.plan_attn_items([(0, 4900)], 0) == [(q, min(128, 4900 - q), 0, 4900) for q in range(0, 4864, 128)] + \
        [(4864, 36, 0, 4900)]


def test_attention_key_split_planner():
    """plan_attn_items_split (host): whole items cover their rows once with the segment as key range; a split pair covers
    the same 128 rows twice with the two halves of the key range meeting at a multiple of 64; pair ids are 0..n-1; the
    rule is per segment (a stacked launch splits the same blocks of each image); short segments are never split."""
    from vision_inspection_system_amd import hip
    items, n = hip.plan_attn_items_split([(0, 4900)], 16)
    assert n == 9 and len(items) == 48 and len(items) * 16 == hip.ATTN_SLOTS
    for segs, heads in [([(0, 4900)], 16), ([(0, 4900), (4928, 9828)], 16), ([(0, 3200)], 3), ([(0, 6404), (6404, 6432)], 16)]:
        items, n = hip.plan_attn_items_split(segs, heads)
        rows, pairs = {}, {}
        seen_split = False
        for (q0, y, k0, k1) in items:
            qn, fl = y & 0xff, y >> 8
            seg = [sg for sg in segs if sg[0] <= q0 < sg[1]][0]
            assert 0 < qn <= 128 and q0 + qn <= seg[1]
            if fl == 0:
                assert not seen_split, "whole items come first"
                assert (k0, k1) == seg
                for r in range(q0, q0 + qn):
                    assert r not in rows
                    rows[r] = 1
            else:
                seen_split = True
                assert fl & 1 and qn == 128
                pairs.setdefault(fl >> 2, {})[(fl >> 1) & 1] = (q0, k0, k1, seg)
        assert sorted(pairs) == list(range(n))
        for parts in pairs.values():
            (qa, a0, a1, seg), (qb, b0, b1, _) = parts[0], parts[1]
            assert qa == qb and a0 == seg[0] and b1 == seg[1] and a1 == b0 and a1 % 64 == 0 and a0 < a1 < b1
            for r in range(qa, qa + 128):
                assert r not in rows
                rows[r] = 1
        assert len(rows) == sum(e - s for s, e in segs)
    one, n1 = hip.plan_attn_items_split([(0, 4900)], 16)
    two, n2 = hip.plan_attn_items_split([(0, 4900), (4928, 9828)], 16)
    assert n2 == 2 * n1
    assert {(q0, y & 0xff, y >> 8 & 3, k0, k1) for (q0, y, k0, k1) in one} <= {(q0, y & 0xff, y >> 8 & 3, k0, k1) for (q0, y, k0, k1) in two}
    assert hip.plan_attn_items_split([(i * 64, (i + 1) * 64) for i in range(80)], 16)[1] == 0
    assert hip.plan_attn_items_split([(0, 2000)], 16)[1] == 0


def test_attention_key_split_entry_refuses_bad_arguments_without_gpu(lib_path):
    """vis_attn_split_ws_bytes is host arithmetic; vis_attn_prefill_split checks head_dim, workspace pointer, alignment and
    size before any HIP call."""
    from vision_inspection_system_amd import hip
    lib = hip.load()
    need = lib.vis_attn_split_ws_bytes(9, 16)
    assert need == 256 * ((9 * 16 * 4 + 255) // 256) + 9 * 16 * 2 * 128 * 84 * 4
    assert lib.vis_attn_split_ws_bytes(0, 16) == 0 and lib.vis_attn_split_ws_bytes(9, 0) == 0
    assert lib.vis_attn_split_ws_bytes(1 << 20, 64) == 0            # more than an int holds
    P = 4096                                                        # never dereferenced: every call below fails a check first
    args = lambda HD=80, n_pairs=9, ws=P, ws_bytes=need: (P, P, P, P, P, 48, 16, 16, HD, 4900, 4900, 4928, 1280, 0.11,
                                                            n_pairs, ws, ws_bytes, None)
    assert lib.vis_attn_prefill_split(*args(HD=128)) == 1
    assert lib.vis_attn_prefill_split(*args(n_pairs=0)) == 1
    assert lib.vis_attn_prefill_split(*args(ws=None)) == 1
    assert lib.vis_attn_prefill_split(*args(ws=P + 16)) == 1        # 256-byte alignment
    assert lib.vis_attn_prefill_split(*args(ws_bytes=need - 1)) == 1


def test_attn_plan_key_ranges_and_whole_rounds():
    """plan_attn_items_split (host), r04 generalisation: a segment may carry its own key range (mllama tower: present rows
    over every canvas row, pad rows over the present ones) and a segment longer than one round of slots is topped up to the
    next whole round; the rule looks at one segment only (same items alone and stacked, shifted by the canvas offset)."""
    from vision_inspection_system_amd import hip
    one, n1 = hip.plan_attn_items_split([(0, 6404, 0, 6432), (6404, 6432, 0, 6404)], 16)
    assert n1 == 45 and len(one) == 6 + 90 + 1              # 51 blocks on 48 slots per head -> two rounds of 48 (+ the pad item)
    halves = [it for it in one if it[1] >> 8]
    assert {(it[2], it[3]) for it in halves} == {(0, 3200), (3200, 6432)}          # key halves cut on a 64-key boundary
    whole = [it for it in one if not it[1] >> 8]
    assert (6404, 28, 0, 6404) in whole and all(it[2:] == (0, 6432) for it in whole if it[0] < 6404)
    two, n2 = hip.plan_attn_items_split([(0, 6404, 0, 6432), (6404, 6432, 0, 6404),
                                         (6464, 6464 + 6404, 6464, 6464 + 6432), (6464 + 6404, 6464 + 6432, 6464, 6464 + 6404)], 16)
    assert n2 == 2 * n1
    first = [it for it in two if it[0] < 6464]
    assert sorted((it[0], it[1] & 0xFF, it[1] >> 8 & 3, it[2], it[3]) for it in first) == \
        sorted((it[0], it[1] & 0xFF, it[1] >> 8 & 3, it[2], it[3]) for it in one)
