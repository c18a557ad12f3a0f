"""Per-kernel numerics: every C-ABI entry point vs a plain PyTorch fp32 reference of the same op.

Tolerances (stated per test): inputs are bf16, accumulation is f32, outputs are
rounded to bf16 once, so the bound is a few bf16 ulps of the output magnitude.
"""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from vision_inspection_system_amd import hip as h
    h.load()
    return h


def _randn(shape, device, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(device)


def _assert_close(got, ref, atol, rtol, what):
    got = got.float().cpu()
    ref = ref.float().cpu()
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = err > bound
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; first at {idx}: "
            f"got {got[tuple(idx)].item():.6f} ref {ref[tuple(idx)].item():.6f}; max err {err.max().item():.6f}")


# ----------------------------------------------------------------------------- K2 GEMM
GEMM_SHAPES = [
    (128, 128, 64), (256, 256, 128), (100, 72, 64), (333, 960, 320), (1, 128, 64),
    (2249, 512, 256), (130, 4608, 3584), (257, 1280, 5120),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_plain(hip, device, M, N, K):
    a = _randn((M, K), device, 1)
    w = _randn((N, K), device, 2, 1.0 / math.sqrt(K))
    out = hip.gemm(a, w)
    ref = a.float() @ w.float().t()
    _assert_close(out, ref, atol=2e-2, rtol=1e-2, what=f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K,kind", [
    (2249, 4608, 3584, "bias"),        # LLM qkv       -> 256x192 tiles (one round)
    (2249, 3584, 3584, "residual"),    # LLM o         -> 256x128 pipelined tiles
    (2249, 37888, 3584, "swiglu"),     # LLM gate/up   -> 256x256 whole rounds + 128x128 remainder columns
    (4900, 5120, 1280, "quickgelu"),   # ViT fc1       -> 256x256
    (4900, 1280, 5120, "residual"),    # ViT fc2       -> 256x128
    (4900, 3840, 1280, "bias"),        # ViT qkv       -> 128x128
    (2300, 768, 1024, "bias"),         # ragged M and N tails on the 256-row kernels
    (2049, 12296, 1088, "bias"),       # ping-pong 256x256: odd K-tile count (17), ragged M and N edges
    (4096, 6144, 1024, "residual"),    # ping-pong 256x256: exactly full tiles, even K-tile count
])
def test_gemm_production_shapes(hip, device, M, N, K, kind):
    """The shapes the 7B prefill actually runs, so that every tile kernel and dispatch branch is parity-checked."""
    from vision_inspection_system_amd.weights import interleave_gate_up
    a = _randn((M, K), device, 11)
    w = _randn((N, K), device, 12, 1.0 / math.sqrt(K))
    af, wf = a.float(), w.float()
    if kind == "swiglu":
        I = N // 2
        out = hip.gemm(a, interleave_gate_up(w[:I].contiguous(), w[I:].contiguous()), act=hip.ACT_SWIGLU)
        ref = torch.nn.functional.silu(af @ wf[:I].t()) * (af @ wf[I:].t())
    elif kind == "residual":
        r = _randn((M, N), device, 13)
        out = hip.gemm(a, w, residual=r)
        ref = af @ wf.t() + r.float()
    elif kind == "quickgelu":
        b = _randn((N,), device, 14)
        out = hip.gemm(a, w, bias=b, act=hip.ACT_QUICKGELU)
        y = af @ wf.t() + b.float()
        ref = y * torch.sigmoid(1.702 * y)
    else:
        b = _randn((N,), device, 14)
        out = hip.gemm(a, w, bias=b)
        ref = af @ wf.t() + b.float()
    _assert_close(out, ref, atol=3e-2, rtol=1e-2, what=f"gemm {M}x{N}x{K} {kind}")


@pytest.mark.parametrize("K", [1024, 1088, 1152, 2048 + 64])
def test_gemm_pingpong_exact_integers(hip, device, K):
    """Small-integer operands: every product and partial sum is exact in f32 and the result is exact in bf16, so a
    single stale or early LDS half-tile (a staging race in the ping-pong schedule) shows as a wrong integer.  Run
    several times: a race would come and go."""
    M, N = 2048 + 40, 12288 + 24          # 9 x 49 tiles of 256 x 256 -> the ping-pong kernel, ragged edges
    g = torch.Generator(device="cpu").manual_seed(K)
    a = torch.randint(-2, 3, (M, K), generator=g).to(torch.bfloat16).to(device)
    w = torch.randint(-1, 2, (N, K), generator=g).to(torch.bfloat16).to(device)
    # keep |sum| < 256 so that the bf16 output is exact: zero out all but 120 columns of k per row of w
    keep = torch.zeros(K, dtype=torch.bool)
    keep[torch.randperm(K, generator=g)[:120]] = True
    w = w * keep.to(device).to(torch.bfloat16)
    ref = (a.float() @ w.float().t())
    assert float(ref.abs().max()) <= 256
    for _ in range(5):
        out = hip.gemm(a, w)
        assert torch.equal(out.float(), ref)


def test_gemm_asymmetric_layout(hip, device):
    # A = I, asymmetric W: catches a transposed accumulator map
    K = 128
    a = torch.eye(K, dtype=torch.bfloat16, device=device)
    w = (torch.arange(256 * K, device=device).reshape(256, K) % 251).to(torch.bfloat16)
    out = hip.gemm(a, w)
    assert torch.equal(out.float(), w.float().t())


@pytest.mark.parametrize("act", ["quickgelu", "gelu"])
def test_gemm_bias_act(hip, device, act):
    M, N, K = 300, 640, 320
    a = _randn((M, K), device, 3)
    w = _randn((N, K), device, 4, 1.0 / math.sqrt(K))
    b = _randn((N,), device, 5)
    code = hip.ACT_QUICKGELU if act == "quickgelu" else hip.ACT_GELU_ERF
    out = hip.gemm(a, w, bias=b, act=code)
    x = a.float() @ w.float().t() + b.float()
    ref = x * torch.sigmoid(1.702 * x) if act == "quickgelu" else torch.nn.functional.gelu(x)
    _assert_close(out, ref, atol=2e-2, rtol=1e-2, what=f"gemm+bias+{act}")


def test_gemm_bias_residual(hip, device):
    M, N, K = 200, 256, 704
    a = _randn((M, K), device, 6)
    w = _randn((N, K), device, 7, 1.0 / math.sqrt(K))
    b = _randn((N,), device, 8)
    r = _randn((M, N), device, 9)
    out = hip.gemm(a, w, bias=b, residual=r)
    ref = a.float() @ w.float().t() + b.float() + r.float()
    _assert_close(out, ref, atol=3e-2, rtol=1e-2, what="gemm+bias+residual")


def test_gemm_swiglu(hip, device):
    M, K, I = 150, 256, 704
    a = _randn((M, K), device, 10)
    wg = _randn((I, K), device, 11, 1.0 / math.sqrt(K))
    wu = _randn((I, K), device, 12, 1.0 / math.sqrt(K))
    from vision_inspection_system_amd.weights import interleave_gate_up
    wgu = interleave_gate_up(wg, wu)
    out = hip.gemm(a, wgu, act=hip.ACT_SWIGLU)
    g = a.float() @ wg.float().t()
    u = a.float() @ wu.float().t()
    ref = torch.nn.functional.silu(g) * u
    _assert_close(out, ref, atol=2e-2, rtol=1e-2, what="gemm swiglu")


def test_gemm_rejects_bad_k(hip, device):
    a = _randn((16, 40), device, 1)
    w = _randn((16, 40), device, 2)
    with pytest.raises(hip.HipLibraryError):
        hip.gemm(a, w)


# ----------------------------------------------------------------------------- K3 / K5 norms
@pytest.mark.parametrize("rows,N", [(1, 256), (7, 3584), (2249, 3584), (33, 1280)])
def test_rmsnorm(hip, device, rows, N):
    x = _randn((rows, N), device, 20, 3.0)
    w = _randn((N,), device, 21)
    out = hip.rmsnorm(x, w, 1e-6)
    xf = x.float()
    normed = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16).float()
    ref = w.float() * normed
    _assert_close(out, ref, atol=2e-2, rtol=1e-2, what="rmsnorm")


@pytest.mark.parametrize("rows,N", [(5, 320), (4900, 1280), (3, 5120)])
def test_layernorm(hip, device, rows, N):
    x = _randn((rows, N), device, 22, 2.0) + 0.5
    w = _randn((N,), device, 23)
    b = _randn((N,), device, 24)
    out = hip.layernorm(x, w, b, 1e-6)
    ref = torch.nn.functional.layer_norm(x.float(), (N,), w.float(), b.float(), 1e-6)
    _assert_close(out, ref, atol=3e-2, rtol=1e-2, what="layernorm")


# ----------------------------------------------------------------------------- K4 rope/split
def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


@pytest.mark.parametrize("S,Hq,Hkv,HD", [(70, 4, 2, 128), (2249, 28, 4, 128), (100, 4, 4, 80), (4900, 16, 16, 80)])
def test_qkv_rope_split(hip, device, S, Hq, Hkv, HD):
    qkv = _randn((S, (Hq + 2 * Hkv) * HD), device, 30)
    g = torch.Generator().manual_seed(31)
    ang = torch.rand((S, HD // 2), generator=g) * 6.28
    emb = torch.cat((ang, ang), dim=-1)
    cos, sin = emb.cos().to(device), emb.sin().to(device)
    T = S + 5
    pos0 = 3
    ld = ((S + 63) // 64) * 64
    q = torch.zeros((Hq, S, HD), dtype=torch.bfloat16, device=device)
    k = torch.zeros((Hkv, T, HD), dtype=torch.bfloat16, device=device)
    v = torch.zeros((Hkv, T, HD), dtype=torch.bfloat16, device=device)
    vt = torch.full((Hkv, HD, ld), 7.0, dtype=torch.bfloat16, device=device)
    hip.qkv_rope_split(qkv, cos, sin, q, k, v, vt, Hq, Hkv, HD, k_pos0=pos0)
    x = qkv.float().reshape(S, Hq + 2 * Hkv, HD)
    c, s_ = cos[:, None, :], sin[:, None, :]
    qr = x[:, :Hq] * c + _rotate_half(x[:, :Hq]) * s_
    kr = x[:, Hq:Hq + Hkv] * c + _rotate_half(x[:, Hq:Hq + Hkv]) * s_
    vr = x[:, Hq + Hkv:]
    _assert_close(q, qr.permute(1, 0, 2), atol=2e-2, rtol=1e-2, what="rope q")
    _assert_close(k[:, pos0:pos0 + S], kr.permute(1, 0, 2), atol=2e-2, rtol=1e-2, what="rope k")
    assert torch.equal(v[:, pos0:pos0 + S].float().cpu(), vr.permute(1, 0, 2).cpu())
    # V^T columns are in the attention kernel's k-slot order (hip.vt_key_order): column c holds key key_of_col[c]
    key_of_col = hip.vt_key_order(ld).cpu()
    plain = torch.zeros((Hkv, HD, ld))
    plain[:, :, :S] = vr.permute(1, 2, 0).cpu()
    assert torch.equal(vt.float().cpu(), plain[:, :, key_of_col])
    assert float(k[:, :pos0].float().abs().max().cpu()) == 0.0


# ----------------------------------------------------------------------------- K6 / K7 attention
def _attn_ref(q, k, v, segments, causal, scale):
    """q [Hq,S,D], k/v [Hkv,S,D] float -> [S, Hq*D]"""
    Hq, S, D = q.shape
    Hkv = k.shape[0]
    out = torch.zeros((S, Hq, D))
    for (s, e) in segments:
        for h in range(Hq):
            kk, vv = k[h // (Hq // Hkv), s:e], v[h // (Hq // Hkv), s:e]
            sc = (q[h, s:e] @ kk.t()) * scale
            if causal:
                n = e - s
                sc = sc.masked_fill(torch.triu(torch.ones(n, n, dtype=torch.bool), 1), float("-inf"))
            out[s:e, h] = torch.softmax(sc, dim=-1) @ vv
    return out.reshape(S, Hq * D)


ATTN_CASES = [
    # S, Hq, Hkv, HD, causal, segments
    (64, 2, 1, 128, True, None),
    (200, 4, 2, 128, True, None),
    (2249, 28, 4, 128, True, None),
    (150, 2, 2, 128, False, None),
    (100, 2, 2, 80, False, None),
    (4900, 16, 16, 80, False, None),
    (356, 4, 4, 80, False, [(0, 100), (100, 356)]),       # two images, unaligned boundary
    (320, 4, 4, 80, False, [(i * 64, (i + 1) * 64) for i in range(5)]),  # Qwen2.5-VL style windows
]


@pytest.mark.parametrize("S,Hq,Hkv,HD,causal,segments", ATTN_CASES)
def test_attn_prefill(hip, device, S, Hq, Hkv, HD, causal, segments):
    segments = segments or [(0, S)]
    q = _randn((Hq, S, HD), device, 40)
    k = _randn((Hkv, S, HD), device, 41)
    v = _randn((Hkv, S, HD), device, 42)
    ld = ((S + 63) // 64) * 64
    vt = torch.zeros((Hkv, HD, ld), dtype=torch.bfloat16, device=device)
    vt[:, :, :S] = v.permute(0, 2, 1)
    vt = vt[:, :, hip.vt_key_order(ld, device)].contiguous()
    out = torch.zeros((S, Hq * HD), dtype=torch.bfloat16, device=device)
    work = hip.make_attn_work(segments, causal, device, heads=Hq)
    scale = HD ** -0.5
    hip.attn_prefill(q, k, vt, out, work, causal, scale)
    if S > 1024:  # reference on two heads (each with its own kv head) to keep CPU time down
        heads = [0, Hq - 1]
        group = Hq // Hkv
        ref = torch.cat([_attn_ref(q[h:h + 1].float().cpu(), k[h // group:h // group + 1].float().cpu(),
                                   v[h // group:h // group + 1].float().cpu(), segments, causal, scale)
                         for h in heads], dim=1)
        got = torch.cat([out[:, h * HD:(h + 1) * HD] for h in heads], dim=1)
    else:
        ref = _attn_ref(q.float().cpu(), k.float().cpu(), v.float().cpu(), segments, causal, scale)
        got = out
    # P is rounded to bf16 before P*V: error ~ 2^-8 relative on O(1) outputs
    _assert_close(got, ref, atol=2e-2, rtol=2e-2, what=f"attn S={S} HD={HD} causal={causal}")


VIT_CASES = [
    # S, H, segments (None = one)
    (384, 2, None),
    (1000, 4, [(0, 333), (333, 1000)]),                    # two images, unaligned boundary: masks at both tile edges
    (6432, 3, [(0, 6404), (6404, 6432)]),                  # mllama-like: a long segment and a 28-row one
]


@pytest.mark.parametrize("S,H,segments", VIT_CASES)
def test_attn_prefill_d80_wide_kernel_equals_narrow_kernel_inputs(hip, device, S, H, segments):
    """head_dim 80, non-causal runs on the 32x32x16 kernel (attn_vit32_kernel: denominator from the pad rows of the V^T
    image, one 32-row block per wave); more shapes against the fp32 reference, launched twice (the LDS pad rows are set
    per launch) and with 128-row as well as planner items."""
    HD = 80
    segments = segments or [(0, S)]
    q = _randn((H, S, HD), device, 140)
    k = _randn((H, S, HD), device, 141)
    v = _randn((H, S, HD), device, 142)
    ld = ((S + 63) // 64) * 64
    vt = torch.zeros((H, HD, ld), dtype=torch.bfloat16, device=device)
    vt[:, :, :S] = v.permute(0, 2, 1)
    vt = vt[:, :, hip.vt_key_order(ld, device)].contiguous()
    scale = HD ** -0.5
    out = torch.full((S, H * HD), 7.0, dtype=torch.bfloat16, device=device)
    plain = torch.zeros_like(out)
    for _ in range(2):
        hip.attn_prefill(q, k, vt, out, hip.make_attn_work(segments, False, device, heads=H), False, scale)
    hip.attn_prefill(q, k, vt, plain, hip.make_attn_work(segments, False, device, heads=0), False, scale)
    assert torch.equal(out, plain), "a row's result must not depend on how the rows are cut into work items"
    heads = [0, H - 1] if S > 1024 else list(range(H))
    ref = torch.cat([_attn_ref(q[h:h + 1].float().cpu(), k[h:h + 1].float().cpu(), v[h:h + 1].float().cpu(), segments,
                               False, scale) for h in heads], dim=1)
    got = torch.cat([out[:, h * HD:(h + 1) * HD] for h in heads], dim=1)
    _assert_close(got, ref, atol=2e-2, rtol=2e-2, what=f"d80 attention S={S}")


def _vt_of(hip, v, S, device):
    ld = ((S + 63) // 64) * 64
    vt = torch.zeros((v.shape[0], v.shape[2], ld), dtype=torch.bfloat16, device=device)
    vt[:, :, :S] = v.permute(0, 2, 1)
    return vt[:, :, hip.vt_key_order(ld, device)].contiguous()


@pytest.mark.parametrize("S,H", [(3200, 3), (4900, 16)])
def test_attn_vit_key_split_plan(hip, device, S, H):
    """Key-split items (vis_attn_prefill_split): two workgroups see half of the keys each and the later one merges.
    Against the fp32 reference and the unsplit launch; launch-to-launch bit-identical (the merge does not depend on which
    half arrives last); the workspace counters are left at zero; a spike in the second half moves the reference there."""
    HD = 80
    q = _randn((H, S, HD), device, 160)
    k = _randn((H, S, HD), device, 161)
    v = _randn((H, S, HD), device, 162)
    k[:, S - 70] = q[:, S - 5] * 4.0                 # a dominating key in the second half for a row of a split block
    vt = _vt_of(hip, v, S, device)
    scale = HD ** -0.5
    plan = hip.make_vit_attn_plan([(0, S)], device, H)
    assert plan.n_pairs > 0 and plan.work.shape[0] * H <= 768
    outs = []
    for i in range(24):                  # back to back on one workspace: a visibility race would show as a differing launch
        out = torch.full((S, H * HD), 3.0, dtype=torch.bfloat16, device=device)
        hip.attn_prefill_plan(q, k, vt, out, plan, scale)
        outs.append(out)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:]), "key-split merge must be deterministic"
    n_count = plan.n_pairs * H
    assert int(plan.ws[:n_count * 4].view(torch.int32).abs().sum()) == 0, "counters must return to zero"
    # two streams launching the same plan at the same time: each has its own workspace
    cur = torch.cuda.current_stream(device)
    side = [torch.cuda.Stream(device=device) for _ in range(2)]
    side_out = [torch.full((S, H * HD), 5.0, dtype=torch.bfloat16, device=device) for _ in side]
    for st, o in zip(side, side_out):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            for _ in range(6):
                hip.attn_prefill_plan(q, k, vt, o, plan, scale)
    for st in side:
        cur.wait_stream(st)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in side_out), "concurrent launches of one plan on two streams"
    plain = torch.zeros_like(outs[0])
    hip.attn_prefill(q, k, vt, plain, hip.make_attn_work([(0, S)], False, device, heads=H), False, scale)
    whole_rows = S - plan.n_pairs * 128 - (S % 128)
    assert torch.equal(outs[0][:whole_rows], plain[:whole_rows]), "unsplit blocks are untouched by the plan"
    _assert_close(outs[0], plain.float(), atol=1e-2, rtol=1e-2, what="key-split vs one pass over all keys")
    heads = [0, H - 1]
    ref = torch.cat([_attn_ref(q[h:h + 1].float().cpu(), k[h:h + 1].float().cpu(), v[h:h + 1].float().cpu(), [(0, S)],
                               False, scale) for h in heads], dim=1)
    got = torch.cat([outs[0][:, h * HD:(h + 1) * HD] for h in heads], dim=1)
    _assert_close(got, ref, atol=2e-2, rtol=2e-2, what=f"key-split attention S={S}")


def test_attn_vit_key_split_is_per_segment(hip, device):
    """The split rule looks at one segment only: an image's rows are bit-identical alone and stacked behind another."""
    S, H, HD = 3136, 2, 80
    scale = HD ** -0.5
    q = _randn((H, 2 * S, HD), device, 170)
    k = _randn((H, 2 * S, HD), device, 171)
    v = _randn((H, 2 * S, HD), device, 172)
    both = torch.zeros((2 * S, H * HD), dtype=torch.bfloat16, device=device)
    plan2 = hip.make_vit_attn_plan([(0, S), (S, 2 * S)], device, H)
    assert plan2.n_pairs == 2 * hip.make_vit_attn_plan([(0, S)], device, H).n_pairs > 0
    hip.attn_prefill_plan(q, k, _vt_of(hip, v, 2 * S, device), both, plan2, scale)
    for i in range(2):
        qi, ki, vi = (t[:, i * S:(i + 1) * S].contiguous() for t in (q, k, v))
        one = torch.zeros((S, H * HD), dtype=torch.bfloat16, device=device)
        hip.attn_prefill_plan(qi, ki, _vt_of(hip, vi, S, device), one, hip.make_vit_attn_plan([(0, S)], device, H), scale)
        assert torch.equal(one, both[i * S:(i + 1) * S]), f"segment {i}: stacked and single launches differ"


def test_attn_prefill_d80_spiked_max_and_cross_keys(hip, device):
    """Rescale branch (a dominating key in a late tile) and keys that are not the queries' own rows (the mllama tower's
    second item group: queries n_real.. attend keys 0..n_real)."""
    S, H, HD = 900, 2, 80
    q = _randn((H, S, HD), device, 150, 0.5)
    k = _randn((H, S, HD), device, 151, 0.5)
    v = _randn((H, S, HD), device, 152)
    k[:, 770] = q[:, 10] * 8.0          # key 770 (13th tile) dominates query 10
    ld = 960
    vt = torch.zeros((H, HD, ld), dtype=torch.bfloat16, device=device)
    vt[:, :, :S] = v.permute(0, 2, 1)
    vt = vt[:, :, hip.vt_key_order(ld, device)].contiguous()
    items = [(q0, min(128, 700 - q0), 0, 900) for q0 in range(0, 700, 128)] + [(700, 128, 0, 700), (828, 72, 0, 700)]
    work = torch.tensor(items, dtype=torch.int32, device=device).reshape(-1, 4).contiguous()
    out = torch.zeros((S, H * HD), dtype=torch.bfloat16, device=device)
    hip.attn_prefill(q, k, vt, out, work, False, HD ** -0.5)
    qf, kf, vf = q.float().cpu(), k.float().cpu(), v.float().cpu()
    ref = torch.empty((S, H, HD))
    for (q0, qn, k0, k1) in items:
        sc = torch.einsum("hqd,hkd->hqk", qf[:, q0:q0 + qn], kf[:, k0:k1]) * HD ** -0.5
        ref[q0:q0 + qn] = torch.einsum("hqk,hkd->qhd", torch.softmax(sc, dim=-1), vf[:, k0:k1])
    _assert_close(out, ref.reshape(S, H * HD), atol=2e-2, rtol=2e-2, what="d80 attention, spiked max / cross keys")


def test_attn_prefill_spiked_max(hip, device):
    """Force the online-softmax rescale branch: one key with a huge score late in the sequence."""
    S, H, HD = 300, 2, 128
    q = _randn((H, S, HD), device, 50, 0.5)
    k = _randn((H, S, HD), device, 51, 0.5)
    v = _randn((H, S, HD), device, 52)
    k[:, 257] = q[:, 299] * 8.0  # key 257 dominates for query 299 (third KV tile)
    ld = 320
    vt = torch.zeros((H, HD, ld), dtype=torch.bfloat16, device=device)
    vt[:, :, :S] = v.permute(0, 2, 1)
    vt = vt[:, :, hip.vt_key_order(ld, device)].contiguous()
    out = torch.zeros((S, H * HD), dtype=torch.bfloat16, device=device)
    work = hip.make_attn_work([(0, S)], True, device)
    hip.attn_prefill(q, k, vt, out, work, True, HD ** -0.5)
    ref = _attn_ref(q.float().cpu(), k.float().cpu(), v.float().cpu(), [(0, S)], True, HD ** -0.5)
    _assert_close(out, ref, atol=2e-2, rtol=2e-2, what="attn spiked max")


@pytest.mark.parametrize("M,N,K,ks", [(300, 520, 1024, 2), (2249, 3584, 18944, 2), (700, 256, 768, 3)])
def test_gemm_splitk(hip, device, M, N, K, ks):
    a = _randn((M, K), device, 30)
    w = _randn((N, K), device, 31, 1.0 / math.sqrt(K))
    b = _randn((N,), device, 32)
    r = _randn((M, N), device, 33)
    work = torch.empty(ks * M * N, dtype=torch.float32, device=device)
    out = hip.gemm_splitk(a, w, work, ks, bias=b, residual=r)
    ref = a.float() @ w.float().t() + b.float() + r.float()
    _assert_close(out, ref, atol=4e-2, rtol=1e-2, what="split-K gemm")
    assert torch.equal(out, hip.gemm_splitk(a, w, work, ks, bias=b, residual=r))     # fixed order: reproducible


# ----------------------------------------------------------------------------- fp8 MFMA GEMM + activation quantiser
@pytest.mark.parametrize("M,K,norm", [(37, 256, True), (300, 3584, True), (300, 3584, False), (64, 18944, False), (101, 5120, False),
                                      (50, 10240, False), (33, 1280, False), (7, 24576, False)])
def test_quant_rows_fp8(hip, device, M, K, norm):
    x = _randn((M, K), device, 170, 2.0)
    nw = _randn((K,), device, 171) if norm else None
    q, sc = hip.quant_rows_fp8(x, norm_w=nw, eps=1e-6)
    xf = x.float()
    if norm:
        xf = ((xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16).float() * nw.float()
              ).to(torch.bfloat16).float()
    ref_sc = (xf.abs().amax(dim=1) / 448.0).clamp_min(1e-12)
    assert torch.allclose(sc, ref_sc, rtol=2e-2 if norm else 1e-6, atol=0)     # norm: rstd differs in the last bits
    ref_q = (xf * (1.0 / sc)[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)   # the kernel multiplies by 1/scale
    mism = (q != ref_q).float().mean().item()
    assert mism < (2e-2 if norm else 1e-6), f"{mism:.4f} of the bytes differ"        # exact without the norm
    deq = q.view(torch.float8_e4m3fn).float() * sc[:, None]
    assert (deq - xf).abs().max() <= xf.abs().amax() / 16 + 1e-6


def test_quant_rows_fp8_fused_layernorm(hip, device):
    M, K = 300, 1280
    x = _randn((M, K), device, 185, 2.0)
    nw, nb = _randn((K,), device, 186), _randn((K,), device, 187)
    q, sc = hip.quant_rows_fp8(x, norm_w=nw, norm_b=nb, eps=1e-6)
    ref = hip.layernorm(x, nw, nb, 1e-6).float()                      # the bf16 values the norm kernel writes
    ref_sc = (ref.abs().amax(dim=1) / 448.0).clamp_min(1e-12)
    assert torch.allclose(sc, ref_sc, rtol=1e-6, atol=0)
    ref_q = (ref * (1.0 / sc)[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    assert (q != ref_q).float().mean().item() < 1e-6


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 520, 1024), (2249, 4608, 3584), (2249, 3584, 18944), (77, 100, 256),
                                   (2249, 37888, 3584), (4900, 1280, 5120), (4900, 3840, 1280)])
def test_gemm_fp8(hip, device, M, N, K):
    """Against the SAME quantised operands in fp32: the kernel adds only f32 accumulation-order effects."""
    a = _randn((M, K), device, 172)
    w = _randn((N, K), device, 173, 1.0 / math.sqrt(K))
    b = _randn((N,), device, 174)
    r = _randn((M, N), device, 175)
    aq, sa = hip.quant_rows_fp8(a)
    wq, sw = hip.quantize_fp8_rows(w)
    out = hip.gemm_fp8(aq, sa, wq, sw, bias=b, residual=r)
    ref = (aq.view(torch.float8_e4m3fn).float() @ wq.view(torch.float8_e4m3fn).float().t()) * sa[:, None] * sw[None, :] \
        + b.float() + r.float()
    _assert_close(out, ref, atol=3e-2, rtol=1e-2, what=f"gemm fp8 {M}x{N}x{K}")
    if N % 8 == 0 and K >= 512:          # split-K form: same result up to the f32 summation order
        work = torch.empty(2 * M * N, dtype=torch.float32, device=device)
        o2 = hip.gemm_fp8(aq, sa, wq, sw, bias=b, residual=r, work=work, ksplit=2)
        _assert_close(o2, ref, atol=3e-2, rtol=1e-2, what=f"gemm fp8 split-K {M}x{N}x{K}")
    # and the quantised product stays close to the bf16 one (fp8 noise ~ 2^-4 / sqrt(K) per term)
    full = a.float() @ w.float().t() + b.float() + r.float()
    assert (out.float() - full).abs().max() < 0.25


@pytest.mark.parametrize("K", [1024, 1152, 2176])
def test_gemm_fp8_pingpong_exact_integers(hip, device, K):
    """Integer-valued e4m3 operands with unit scales: the fp8 ping-pong tile kernel must reproduce the integer product
    exactly, run after run (8, 9 and 17 K-tiles: even and odd; ragged M and N edges)."""
    M, N = 2048 + 40, 12288 + 24
    g = torch.Generator(device="cpu").manual_seed(K)
    a = torch.randint(-2, 3, (M, K), generator=g).float()
    w = torch.randint(-1, 2, (N, K), generator=g).float()
    keep = torch.zeros(K)
    keep[torch.randperm(K, generator=g)[:120]] = 1.0
    w = w * keep
    aq = a.to(torch.float8_e4m3fn).view(torch.uint8).to(device)
    wq = w.to(torch.float8_e4m3fn).view(torch.uint8).to(device)
    sa = torch.ones(M, dtype=torch.float32, device=device)
    sw = torch.ones(N, dtype=torch.float32, device=device)
    ref = (a @ w.t()).to(device)
    assert float(ref.abs().max()) <= 256
    for _ in range(5):
        out = hip.gemm_fp8(aq, sa, wq, sw)
        assert torch.equal(out.float(), ref)


def test_gemm_fp8_swiglu_and_asymmetric_identity(hip, device):
    from vision_inspection_system_amd.weights import interleave_gate_up
    M, K, I = 200, 256, 704
    a = _randn((M, K), device, 176)
    wg = _randn((I, K), device, 177, 1.0 / math.sqrt(K))
    wu = _randn((I, K), device, 178, 1.0 / math.sqrt(K))
    aq, sa = hip.quant_rows_fp8(a)
    wq, sw = hip.quantize_fp8_rows(interleave_gate_up(wg, wu))
    out = hip.gemm_fp8(aq, sa, wq, sw, act=hip.ACT_SWIGLU)
    d = (wq.view(torch.float8_e4m3fn).float() * sw[:, None]).view(I // 16, 2, 16, K)
    af = aq.view(torch.float8_e4m3fn).float() * sa[:, None]
    ref = torch.nn.functional.silu(af @ d[:, 0].reshape(I, K).t()) * (af @ d[:, 1].reshape(I, K).t())
    _assert_close(out, ref, atol=2e-2, rtol=1e-2, what="gemm fp8 swiglu")
    # A = I (exact in e4m3) with an asymmetric W: catches row/col swaps in the C write
    n = 256
    eye = torch.eye(n, device=device).to(torch.bfloat16)
    w2 = (torch.arange(n * n, device=device).reshape(n, n) % 15 - 7).to(torch.bfloat16)   # row amax 7 -> scale 1/64: exact in e4m3
    eq, es = hip.quant_rows_fp8(eye)
    w2q, w2s = hip.quantize_fp8_rows(w2)
    o2 = hip.gemm_fp8(eq, es, w2q, w2s)
    _assert_close(o2, w2.float().t(), atol=1e-2, rtol=1e-2, what="fp8 identity")


@pytest.mark.parametrize("B,N,K", [(1, 512, 256), (5, 1000, 768), (16, 4608, 3584), (8, 3584, 18944), (19, 1000, 768),
                                   (32, 4608, 3584), (40, 1000, 768), (64, 4608, 3584)])
def test_decode_gemm_fp8_and_finalize(hip, device, B, N, K):
    """fp8 batched-decode projection + fp8 finalisation against fp32 arithmetic on the same quantised operands."""
    x = _randn((B, K), device, 180, 2.0)
    w = _randn((N, K), device, 181, 1.0 / math.sqrt(K))
    b = _randn((N,), device, 182)
    r = _randn((B, N), device, 183)
    nw = _randn((N,), device, 184)
    xq, sx = hip.quant_rows_fp8(x)
    wq, sw = hip.quantize_fp8_rows(w)
    part = torch.full((16 * hip.part_rows(B) * N,), float("nan"), dtype=torch.float32, device=device)
    y = torch.empty((B, N), dtype=torch.bfloat16, device=device)
    yn = torch.empty((B, N), dtype=torch.bfloat16, device=device)
    yq = torch.zeros((B, N), dtype=torch.uint8, device=device)
    ysc = torch.empty(B, dtype=torch.float32, device=device)
    ks = hip.decode_gemm_fp8(xq, sx, wq, sw, part=part)
    assert 1 <= ks <= 16
    hip.skinny_finalize_fp8(part, ks, y, N, sx=sx, sw=sw, bias=b, residual=r, norm_w=nw, yn=yn, yq=yq, yq_scale=ysc)
    prod = (xq.view(torch.float8_e4m3fn).float() @ wq.view(torch.float8_e4m3fn).float().t()) * sx[:, None] * sw[None, :]
    ref = prod + b.float() + r.float()
    _assert_close(y, ref, atol=4e-2, rtol=1e-2, what=f"decode gemm fp8 {B}x{N}x{K}")
    ynf = yn.float()
    ref_sc = (ynf.abs().amax(dim=1) / 448.0).clamp_min(1e-12)
    assert torch.allclose(ysc, ref_sc, rtol=1e-6, atol=0)
    ref_q = (ynf * (1.0 / ysc)[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    assert (yq != ref_q).float().mean().item() < 1e-6           # the fused quantiser equals the stand-alone one
    # direct (scaled) outputs
    out = torch.empty((B, N), dtype=torch.float32, device=device)
    hip.decode_gemm_fp8(xq, sx, wq, sw, out=out)
    _assert_close(out, prod, atol=3e-2, rtol=1e-2, what="decode gemm fp8 direct")


def test_attn_prefill_row_offset_equals_full_pass(hip, device):
    """vis_attn_prefill_rows: the causal pass over rows P.. of a sequence (Q / O hold only those rows, keys 0.. come from
    the cache) gives bit-identical rows to the pass over the whole sequence."""
    Hq, Hkv, HD, S, P = 8, 2, 128, 700, 320
    q = _randn((Hq, S, HD), device, 300)
    k = _randn((Hkv, S, HD), device, 301)
    v = _randn((Hkv, S, HD), device, 302)
    ld = (S + 63) // 64 * 64
    vt = torch.zeros((Hkv, HD, ld), dtype=torch.bfloat16, device=device)
    vt[:, :, :S] = v.transpose(1, 2)
    vt = vt[:, :, hip.vt_key_order(ld, device)].contiguous()
    full = torch.empty((S, Hq * HD), dtype=torch.bfloat16, device=device)
    hip.attn_prefill(q, k, vt, full, hip.make_attn_work([(0, S)], True, device), True, HD ** -0.5)
    items = sorted([(q0, min(128, S - q0), 0, S) for q0 in range(P, S, 128)], key=lambda it: -(it[0] + it[1]))
    work = torch.tensor(items, dtype=torch.int32, device=device).reshape(-1, 4).contiguous()
    part = torch.empty((S - P, Hq * HD), dtype=torch.bfloat16, device=device)
    hip.attn_prefill(q[:, P:].contiguous(), k, vt, part, work, True, HD ** -0.5, q_row0=P)
    assert torch.equal(part, full[P:])
    # and against fp32 arithmetic
    qf, kf, vf = q.float(), k.float().repeat_interleave(Hq // Hkv, 0), v.float().repeat_interleave(Hq // Hkv, 0)
    sc = torch.einsum("hqd,hkd->hqk", qf[:, P:], kf) * HD ** -0.5
    mask = torch.arange(S, device=device)[None, :] <= (torch.arange(P, S, device=device)[:, None])
    ref = torch.einsum("hqk,hkd->qhd", torch.softmax(sc.masked_fill(~mask[None], float("-inf")), -1), vf).reshape(S - P, Hq * HD)
    _assert_close(part, ref, atol=2e-2, rtol=2e-2, what="attention with row offset")


# ----------------------------------------------------------------------------- K10 GEMV
@pytest.mark.parametrize("N,K", [(512, 256), (4608, 3584), (3584, 18944), (1000, 704)])
def test_gemv_plain(hip, device, N, K):
    x = _randn((K,), device, 60)
    w = _randn((N, K), device, 61, 1.0 / math.sqrt(K))
    b = _randn((N,), device, 62)
    r = _randn((N,), device, 63)
    out = torch.empty((N,), dtype=torch.bfloat16, device=device)
    hip.gemv(x, w, out, bias=b, residual=r)
    ref = w.float() @ x.float() + b.float() + r.float()
    _assert_close(out, ref, atol=3e-2, rtol=1e-2, what="gemv")


def test_gemv_fused_rmsnorm_f32_out(hip, device):
    N, K = 1536, 3584
    x = _randn((K,), device, 64, 2.0)
    nw = _randn((K,), device, 65)
    w = _randn((N, K), device, 66, 1.0 / math.sqrt(K))
    out = torch.empty((N,), dtype=torch.float32, device=device)
    hip.gemv(x, w, out, norm_w=nw, eps=1e-6)
    xf = x.float()
    xn = (xf * torch.rsqrt(xf.pow(2).mean() + 1e-6)).to(torch.bfloat16).float() * nw.float()
    xn = xn.to(torch.bfloat16).float()
    ref = w.float() @ xn
    _assert_close(out, ref, atol=2e-2, rtol=1e-2, what="gemv fused rmsnorm")


def test_gemv_swiglu(hip, device):
    K, I = 256, 704
    x = _randn((K,), device, 67)
    wg = _randn((I, K), device, 68, 1.0 / math.sqrt(K))
    wu = _randn((I, K), device, 69, 1.0 / math.sqrt(K))
    from vision_inspection_system_amd.weights import interleave_gate_up
    wgu = interleave_gate_up(wg, wu)
    out = torch.empty((I,), dtype=torch.bfloat16, device=device)
    hip.gemv(x, wgu, out, act=hip.ACT_SWIGLU)
    ref = torch.nn.functional.silu(wg.float() @ x.float()) * (wu.float() @ x.float())
    _assert_close(out, ref, atol=2e-2, rtol=1e-2, what="gemv swiglu")


# ----------------------------------------------------------------------------- K10 with fp8 weights (configs[4] slice)
def _dequant(wq, scale):
    return wq.view(torch.float8_e4m3fn).float() * scale[:, None]


@pytest.mark.parametrize("N,K", [(512, 256), (4608, 3584), (3584, 18944), (1002, 704), (152064, 3584)])
def test_gemv_fp8_weights(hip, device, N, K):
    """Against the SAME quantised weights dequantised in fp32: the kernel's only rounding is bf16 x and f32 sums."""
    x = _randn((K,), device, 160)
    w = _randn((N, K), device, 161, 1.0 / math.sqrt(K))
    b = _randn((N,), device, 162)
    r = _randn((N,), device, 163)
    wq, sc = hip.quantize_fp8_rows(w)
    assert wq.dtype == torch.uint8 and wq.shape == (N, K) and sc.shape == (N,)
    out = torch.empty((N,), dtype=torch.bfloat16, device=device)
    hip.gemv_fp8(x, wq, sc, out, bias=b, residual=r)
    ref = _dequant(wq, sc) @ x.float() + b.float() + r.float()
    _assert_close(out, ref, atol=3e-2, rtol=1e-2, what="gemv fp8")
    # and the quantisation itself stays within e4m3's half-ulp of the bf16 weights (relative 2^-4 of the row max)
    assert ((_dequant(wq, sc) - w.float()).abs().amax(dim=1) <= w.float().abs().amax(dim=1) / 16 + 1e-6).all()


def test_gemv_fp8_fused_rmsnorm_swiglu_f32(hip, device):
    from vision_inspection_system_amd.weights import interleave_gate_up
    K, I = 256, 704
    x = _randn((K,), device, 164, 2.0)
    nw = _randn((K,), device, 165)
    wg = _randn((I, K), device, 166, 1.0 / math.sqrt(K))
    wu = _randn((I, K), device, 167, 1.0 / math.sqrt(K))
    wq, sc = hip.quantize_fp8_rows(interleave_gate_up(wg, wu))
    out = torch.empty((I,), dtype=torch.bfloat16, device=device)
    hip.gemv_fp8(x, wq, sc, out, norm_w=nw, act=hip.ACT_SWIGLU, eps=1e-6)
    xf = x.float()
    xn = ((xf * torch.rsqrt(xf.pow(2).mean() + 1e-6)).to(torch.bfloat16).float() * nw.float()).to(torch.bfloat16).float()
    d = _dequant(wq, sc).view(I // 16, 2, 16, K)
    g, u = d[:, 0].reshape(I, K) @ xn, d[:, 1].reshape(I, K) @ xn
    _assert_close(out, torch.nn.functional.silu(g) * u, atol=2e-2, rtol=1e-2, what="gemv fp8 swiglu")
    N = 1536
    w = _randn((N, K), device, 168, 1.0 / math.sqrt(K))
    wq, sc = hip.quantize_fp8_rows(w)
    o32 = torch.empty((N,), dtype=torch.float32, device=device)
    hip.gemv_fp8(x, wq, sc, o32, norm_w=nw, eps=1e-6)
    _assert_close(o32, _dequant(wq, sc) @ xn, atol=2e-2, rtol=1e-2, what="gemv fp8 f32 out")


# ----------------------------------------------------------------------------- K4/K11 fused decode attention
@pytest.mark.parametrize("Hq,Hkv,ctx0,steps,T", [(2, 1, 37, 3, 128), (28, 4, 2249, 2, 4096), (28, 4, 127, 3, 256)])
def test_decode_attention_fused(hip, device, Hq, Hkv, ctx0, steps, T):
    """rope(q,k) + KV append + attention over the cache, position read from device memory."""
    HD = 128
    kc = torch.zeros((Hkv, T, HD), dtype=torch.bfloat16, device=device)
    vc = torch.zeros((Hkv, T, HD), dtype=torch.bfloat16, device=device)
    kc[:, :ctx0] = _randn((Hkv, ctx0, HD), device, 70)
    vc[:, :ctx0] = _randn((Hkv, ctx0, HD), device, 71)
    g = torch.Generator().manual_seed(72)
    ang = torch.rand((T, HD // 2), generator=g) * 6.28
    emb = torch.cat((ang, ang), -1)
    cos_t, sin_t = emb.cos().to(device), emb.sin().to(device)
    step = torch.full((1,), ctx0, dtype=torch.int32, device=device)
    nsplit = T // hip.DECODE_KEYS_PER_SPLIT
    part_o = torch.empty(Hq * nsplit * HD, dtype=torch.float32, device=device)
    part_ml = torch.empty(Hq * nsplit * 2, dtype=torch.float32, device=device)
    out = torch.empty((Hq * HD,), dtype=torch.bfloat16, device=device)
    kref, vref = kc.float().cpu().clone(), vc.float().cpu().clone()
    for t in range(steps):
        slot = ctx0 + t
        qkv = _randn(((Hq + 2 * Hkv) * HD,), device, 80 + t)
        hip.decode_attn(qkv, cos_t, sin_t, kc, vc, step, part_o, part_ml, out, Hq, Hkv, HD, nsplit, HD ** -0.5)
        x = qkv.float().cpu().reshape(Hq + 2 * Hkv, HD)
        c, s_ = emb[slot].cos(), emb[slot].sin()
        qr = (x[:Hq] * c + _rotate_half(x[:Hq]) * s_).to(torch.bfloat16).float()
        kr = (x[Hq:Hq + Hkv] * c + _rotate_half(x[Hq:Hq + Hkv]) * s_).to(torch.bfloat16).float()
        kref[:, slot] = kr
        vref[:, slot] = x[Hq + Hkv:]
        n = slot + 1
        ref = torch.zeros(Hq, HD)
        for h in range(Hq):
            kv = h // (Hq // Hkv)
            p = torch.softmax((kref[kv, :n] @ qr[h]) * HD ** -0.5, dim=0)
            ref[h] = p @ vref[kv, :n]
        _assert_close(out.reshape(Hq, HD), ref, atol=2e-2, rtol=2e-2, what=f"decode attn step {t}")
        step += 1
    _assert_close(kc[:, :ctx0 + steps], kref[:, :ctx0 + steps], atol=2e-2, rtol=1e-2, what="kv cache k")
    assert torch.equal(vc[:, :ctx0 + steps].float().cpu(), vref[:, :ctx0 + steps])


# ----------------------------------------------------------------------------- K12 glue
def test_argmax_and_step(hip, device):
    V = 152064
    g = torch.Generator().manual_seed(90)
    logits = torch.randn(V, generator=g).to(device)
    logits[77777] = 50.0
    logits[99999] = 50.0  # tie: first index wins, like torch.argmax
    ws_v = torch.empty(256, dtype=torch.float32, device=device)
    ws_i = torch.empty(256, dtype=torch.int32, device=device)
    tokens = torch.full((4,), -1, dtype=torch.int32, device=device)
    cur = torch.zeros(1, dtype=torch.int32, device=device)
    step = torch.zeros(1, dtype=torch.int32, device=device)
    hip.argmax(logits, ws_v, ws_i, tokens, cur, step)
    logits[5] = 60.0
    hip.argmax(logits, ws_v, ws_i, tokens, cur, step)
    assert tokens.cpu().tolist() == [77777, 5, -1, -1]
    assert int(cur.cpu()) == 5 and int(step.cpu()) == 2


def test_gather_scatter_rows(hip, device):
    table = _randn((512, 256), device, 91)
    ids = torch.tensor([5, 0, 511, 17, 5], dtype=torch.int32, device=device)
    out = torch.empty((5, 256), dtype=torch.bfloat16, device=device)
    hip.gather_rows(table, ids, out)
    assert torch.equal(out.cpu(), table.cpu()[ids.cpu().long()])
    dst = torch.zeros((9, 256), dtype=torch.bfloat16, device=device)
    idx = torch.tensor([8, 2, 3], dtype=torch.int32, device=device)
    hip.scatter_rows(out[:3].contiguous(), idx, dst)
    assert torch.equal(dst.cpu()[[8, 2, 3]], out.cpu()[:3])
    assert float(dst.cpu()[[0, 1, 4, 5, 6, 7]].float().abs().max()) == 0.0


def test_patchify_matches_reference_layout(hip, device):
    import numpy as np
    H, W = 56, 84
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mean = (0.48145466, 0.4578275, 0.40821073)
    std = (0.26862954, 0.26130258, 0.27577711)
    out = torch.full((24, 1216), 9.0, dtype=torch.bfloat16, device=device)
    hip.patchify(torch.from_numpy(img).to(device), out, 0, mean, std)
    # numpy restatement of the patch layout (channel-first, 2x2 merge groups, temporal x2)
    x = (img.astype(np.float32) / 255.0 - np.array(mean, np.float32)) / np.array(std, np.float32)
    x = x.transpose(2, 0, 1)
    gh, gw = H // 14, W // 14
    p = x.reshape(3, gh // 2, 2, 14, gw // 2, 2, 14).transpose(1, 4, 2, 5, 0, 3, 6)
    p = np.broadcast_to(p[:, :, :, :, :, None], (*p.shape[:5], 2, 14, 14)).reshape(gh * gw, 1176)
    got = out.float().cpu().numpy()
    assert np.abs(got[:, :1176] - p).max() < 2e-2
    assert np.abs(got[:, 1176:]).max() == 0.0


# ----------------------------------------------------------------------------- K10 batched decode projection
@pytest.mark.parametrize("B,N,K", [(1, 512, 256), (5, 1000, 704), (16, 4608, 3584), (8, 3584, 18944),
                                   (17, 1000, 704), (32, 4608, 3584), (24, 3584, 18944),   # > 16: two 16-row MFMA blocks
                                   (33, 1000, 704), (64, 4608, 3584), (48, 3584, 18944)])  # > 32: four, 64-row x tile
def test_decode_gemm_split_and_finalize(hip, device, B, N, K):
    x = _randn((B, K), device, 100, 2.0)
    w = _randn((N, K), device, 101, 1.0 / math.sqrt(K))
    b = _randn((N,), device, 102)
    r = _randn((B, N), device, 103)
    nw = _randn((N,), device, 104)
    part = torch.empty(16 * hip.part_rows(B) * N, dtype=torch.float32, device=device)
    y = torch.empty((B, N), dtype=torch.bfloat16, device=device)
    yn = torch.empty((B, N), dtype=torch.bfloat16, device=device)
    ks = hip.decode_gemm(x, w, part=part)
    assert 1 <= ks <= 16
    hip.skinny_finalize(part, ks, y, N, bias=b, residual=r, norm_w=nw, yn=yn)
    ref = x.float() @ w.float().t() + b.float() + r.float()
    _assert_close(y, ref, atol=4e-2, rtol=1e-2, what=f"decode gemm {B}x{N}x{K} ks={ks}")
    yf = y.float()
    ref_n = (yf * torch.rsqrt(yf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16).float() * nw.float()
    _assert_close(yn, ref_n, atol=3e-2, rtol=1e-2, what="finalize next-norm")
    y2 = torch.empty_like(y)
    hip.decode_gemm(x, w, part=part, ksplit=ks)
    hip.skinny_finalize(part, ks, y2, N, bias=b, residual=r)
    assert torch.equal(y, y2)            # fixed summation order: bitwise reproducible
    if ks < 16:                          # extra slots are zero-filled, not left as garbage
        part.fill_(float("nan"))
        hip.decode_gemm(x, w, part=part, ksplit=ks + 1)
        hip.skinny_finalize(part, ks + 1, y2, N, bias=b, residual=r)
        assert torch.equal(y, y2)


def test_decode_gemm_swiglu_and_direct_logits(hip, device):
    from vision_inspection_system_amd.weights import interleave_gate_up
    B, K, I = 7, 256, 704
    x = _randn((B, K), device, 105)
    wg = _randn((I, K), device, 106, 1.0 / math.sqrt(K))
    wu = _randn((I, K), device, 107, 1.0 / math.sqrt(K))
    part = torch.empty(16 * 16 * 2 * I, dtype=torch.float32, device=device)
    out = torch.empty((B, I), dtype=torch.bfloat16, device=device)
    ks = hip.decode_gemm(x, interleave_gate_up(wg, wu), part=part)
    hip.skinny_finalize(part, ks, out, 2 * I, swiglu=True)
    ref = torch.nn.functional.silu(x.float() @ wg.float().t()) * (x.float() @ wu.float().t())
    _assert_close(out, ref, atol=3e-2, rtol=1e-2, what="decode gemm swiglu")
    wl = _randn((152064, K), device, 108, 1.0 / math.sqrt(K))
    logits = torch.empty((B, 152064), dtype=torch.float32, device=device)
    hip.decode_gemm(x, wl, out=logits)
    _assert_close(logits, x.float() @ wl.float().t(), atol=2e-2, rtol=1e-2, what="decode gemm f32 logits")
    yb = torch.empty((B, 1000), dtype=torch.bfloat16, device=device)
    hip.decode_gemm(x, wl[:1000], out=yb)
    _assert_close(yb, x.float() @ wl[:1000].float().t(), atol=3e-2, rtol=1e-2, what="decode gemm direct bf16")


@pytest.mark.parametrize("B,N,K,direct", [(5, 524288, 64, False), (40, 262144, 128, True), (20, 300032, 64, False)])
def test_decode_gemm_many_short_tiles_per_workgroup(hip, device, B, N, K, direct):
    """A workgroup's range over more tiles than it has accumulator sets (small K, huge N: 16 / 8 / 9 tiles per workgroup
    against 8 / 6 / 8 sets): the sets past the last one are stored on the spot and reused."""
    x = _randn((B, K), device, 120)
    w = _randn((N, K), device, 121, 1.0 / math.sqrt(K))
    ref = x.float() @ w.float().t()
    if direct:
        out = torch.empty((B, N), dtype=torch.float32, device=device)
        hip.decode_gemm(x, w, out=out)
        _assert_close(out, ref, atol=2e-2, rtol=1e-2, what="direct output over many tiles per workgroup")
        return
    ks = hip.load().vis_gemm_decode_ksplit(N, K)
    part = torch.full((ks * hip.part_rows(B) * N,), float("nan"), dtype=torch.float32, device=device)
    assert hip.decode_gemm(x, w, part=part) == ks
    got = part.view(ks, hip.part_rows(B), N)[:, :B].sum(0)
    _assert_close(got, ref, atol=2e-2, rtol=1e-2, what="partials over many tiles per workgroup")


def test_batched_decode_attention_and_argmax(hip, device):
    """B sequences with different context lengths in one launch == B single-sequence launches."""
    Hq, Hkv, HD, T, B = 28, 4, 128, 512, 3
    ctx = [37, 300, 511]
    kc = _randn((B, Hkv, T, HD), device, 110)
    vc = _randn((B, Hkv, T, HD), device, 111)
    g = torch.Generator().manual_seed(112)
    ang = torch.rand((B, T, HD // 2), generator=g) * 6.28
    emb = torch.cat((ang, ang), -1)
    cos_t, sin_t = emb.cos().to(device), emb.sin().to(device)
    qkv = _randn((B, (Hq + 2 * Hkv) * HD), device, 113)
    step = torch.tensor(ctx, dtype=torch.int32, device=device)
    ns = T // hip.DECODE_KEYS_PER_SPLIT
    po = torch.empty(B * Hq * ns * HD, dtype=torch.float32, device=device)
    pml = torch.empty(B * Hq * ns * 2, dtype=torch.float32, device=device)
    out = torch.empty((B, Hq * HD), dtype=torch.bfloat16, device=device)
    kc1, vc1 = kc.clone(), vc.clone()
    hip.decode_attn(qkv, cos_t, sin_t, kc, vc, step, po, pml, out, Hq, Hkv, HD, ns, HD ** -0.5)
    for b in range(B):
        o1 = torch.empty((Hq * HD,), dtype=torch.bfloat16, device=device)
        hip.decode_attn(qkv[b].contiguous(), cos_t[b].contiguous(), sin_t[b].contiguous(), kc1[b].contiguous(),
                        vc1[b].contiguous(), step[b:b + 1].clone(), po, pml, o1, Hq, Hkv, HD, ns, HD ** -0.5)
        assert torch.equal(out[b], o1), f"sequence {b}"
    # batched argmax: per-sequence logits, tokens, counters
    V = 5000
    lg = torch.randn((B, V), generator=torch.Generator().manual_seed(114)).to(device)
    for b in range(B):
        lg[b, 100 * (b + 1)] = 99.0
    tokens = torch.full((B, 8), -1, dtype=torch.int32, device=device)
    cur = torch.zeros(B, dtype=torch.int32, device=device)
    st = torch.tensor([0, 3, 7], dtype=torch.int32, device=device)
    hip.argmax(lg, torch.empty(256 * B, device=device), torch.empty(256 * B, dtype=torch.int32, device=device),
               tokens, cur, st)
    assert cur.cpu().tolist() == [100, 200, 300] and st.cpu().tolist() == [1, 4, 8]
    t = tokens.cpu()
    assert t[0, 0] == 100 and t[1, 3] == 200 and t[2, 7] == 300 and int((t >= 0).sum()) == 3


@pytest.mark.parametrize("S,Hq,Hkv,P", [(2249, 28, 4, 0), (300, 4, 2, 0), (129, 2, 1, 0), (64, 2, 2, 0), (1000, 8, 2, 448),
                                         (17, 2, 1, 0)])
def test_attn_prefill_pairs_equals_unpaired_and_reference(hip, device, S, Hq, Hkv, P):
    """K7, balanced form: a late and an early 128-row query block per workgroup.  Against the fp32 reference and - bit
    for bit - against the one-block-per-workgroup kernel (same tiles, same order, same rounding points), including an
    odd block count, a ragged last block, a single short block and a shared-prefix pass (rows P.. only)."""
    HD = 128
    q = _randn((Hq, S, HD), device, 70)
    k = _randn((Hkv, S + 5, HD), device, 71)
    v = _randn((Hkv, S + 5, HD), device, 72)
    ld = (S + 63) // 64 * 64
    vt = torch.zeros((Hkv, HD, ld), dtype=torch.bfloat16, device=device)
    vt[:, :, :S] = v[:, :S].transpose(1, 2)
    vt = vt[:, :, hip.vt_key_order(ld, device)].contiguous()
    n = S - P
    qs = q[:, P:].contiguous()
    paired = torch.full((n, Hq * HD), 7.0, dtype=torch.bfloat16, device=device)
    hip.attn_prefill_pairs(qs, k, vt, paired, hip.make_attn_pairs(P, S, device), HD ** -0.5, q_row0=P)
    items = [(q0, min(128, S - q0), 0, S) for q0 in range(P, S, 128)]
    work = torch.tensor(items, dtype=torch.int32, device=device).reshape(-1, 4).contiguous()
    plain = torch.empty_like(paired)
    hip.attn_prefill(qs, k, vt, plain, work, True, HD ** -0.5, q_row0=P)
    assert torch.equal(paired, plain)
    g = Hq // Hkv
    kk, vv = k[:, :S].float().repeat_interleave(g, 0), v[:, :S].float().repeat_interleave(g, 0)
    sc = torch.einsum("hqd,hkd->hqk", q.float(), kk) * HD ** -0.5
    sc = sc.masked_fill(torch.ones(S, S, dtype=torch.bool, device=device).triu(1), float("-inf"))
    ref = torch.einsum("hqk,hkd->qhd", torch.softmax(sc, -1), vv).reshape(S, Hq * HD)[P:]
    _assert_close(paired, ref, atol=2e-2, rtol=2e-2, what="paired causal attention")


@pytest.mark.parametrize("M,N,K,ln", [(2249, 3584, 18944, False), (4900, 1280, 5120, True), (70, 256, 704 - 64, False),
                                       (33, 320, 1280, True)])
def test_splitk_finalize_fused_with_norm_is_bit_identical(hip, device, M, N, K, ln):
    """K-slice partials -> x = sum + bias + residual and y = norm(x) in ONE row pass (vis_splitk_finalize_norm) equals
    vis_gemm_bf16_splitk followed by vis_rmsnorm_bf16 / vis_layernorm_bf16 bit for bit (same slice order, norm statistics
    of the rounded x), with the residual aliasing the output as the engines use it; y = None finalises only."""
    a = _randn((M, K), device, 80)
    w = _randn((N, K), device, 81, scale=K ** -0.5)
    bias = _randn((N,), device, 82) if ln else None
    res = _randn((M, N), device, 83)
    nw, nb = _randn((N,), device, 84), (_randn((N,), device, 85) if ln else None)
    work = torch.empty(2 * M * N, dtype=torch.float32, device=device)
    x_ref = hip.gemm_splitk(a, w, work, 2, bias=bias, residual=res)
    y_ref = hip.layernorm(x_ref, nw, nb, 1e-6) if ln else hip.rmsnorm(x_ref, nw, 1e-6)
    x = res.clone()                                      # residual aliases the output
    y = torch.empty_like(x)
    hip.gemm_splitk_part(a, w, work, 2)
    hip.splitk_finalize_norm(work, 2, x, bias=bias, residual=x, norm_w=nw, norm_b=nb, y_out=y, eps=1e-6)
    assert torch.equal(x, x_ref) and torch.equal(y, y_ref)
    x2 = res.clone()
    hip.splitk_finalize_norm(work, 2, x2, bias=bias, residual=x2)
    assert torch.equal(x2, x_ref)
    ref = a.float() @ w.float().t() + res.float() + (bias.float() if ln else 0.0)
    _assert_close(x, ref, atol=3e-2, rtol=2e-2, what="split-K sum")


@pytest.mark.parametrize("Hq,Hkv,B,T", [(28, 4, 32, 2560), (32, 8, 16, 768), (2, 1, 3, 256)])
def test_decode_attn_streaming_form(hip, device, monkeypatch, Hq, Hkv, B, T):
    """K11, batched streaming form (one workgroup per (kv head, sequence), whole context in one pass, online softmax,
    no partials / combine): ragged context lengths incl. tile boundaries, against an fp32 reference and against the
    split + combine form (same arithmetic per key, other summation order: f32-rounding-level agreement), KV append
    included.  Cross-attention mode (static keys, fused q-norm) the same way."""
    HD = 128
    rng = np.random.default_rng(5)
    ctx = [int(c) for c in rng.integers(1, T - 1, B)]
    ctx[0], ctx[1] = 63, 64
    if B > 2:
        ctx[2] = T - 2
    kc = _randn((B, Hkv, T, HD), device, 120)
    vc = _randn((B, Hkv, T, HD), device, 121)
    g = torch.Generator().manual_seed(122)
    ang = torch.rand((T, HD // 2), generator=g) * 6.28
    emb = torch.cat((ang, ang), -1)
    cos_t = emb.cos().to(device).unsqueeze(0).expand(B, -1, -1)
    sin_t = emb.sin().to(device).unsqueeze(0).expand(B, -1, -1)
    qkv = _randn((B, (Hq + 2 * Hkv) * HD), device, 123)
    step = torch.tensor(ctx, dtype=torch.int32, device=device)
    ns = T // hip.DECODE_KEYS_PER_SPLIT
    nmax = max(ns, -(-448 // hip.DECODE_KEYS_PER_SPLIT))
    po = torch.empty(B * Hq * nmax * HD, dtype=torch.float32, device=device)
    pml = torch.empty(B * Hq * nmax * 2, dtype=torch.float32, device=device)
    outs, caches = {}, {}
    for mode in ("2", "0"):            # 2: streaming form forced, 0: split + combine
        monkeypatch.setenv("VIS_DECODE_ATTN_STREAM", mode)
        k1, v1 = kc.clone(), vc.clone()
        out = torch.full((B, Hq * HD), 7.0, dtype=torch.bfloat16, device=device)
        hip.decode_attn(qkv, cos_t, sin_t, k1, v1, step, po, pml, out, Hq, Hkv, HD, ns, HD ** -0.5)
        outs[mode], caches[mode] = out.float().cpu(), (k1, v1)
    assert torch.equal(caches["2"][0], caches["0"][0]) and torch.equal(caches["2"][1], caches["0"][1])     # same append
    assert (outs["2"] - outs["0"]).abs().max() < 2e-2
    k1, v1 = caches["2"]
    G = Hq // Hkv
    for b in range(B):
        n = ctx[b] + 1
        x = qkv[b].float().cpu().reshape(Hq + 2 * Hkv, HD)
        c, s_ = emb[ctx[b]].cos(), emb[ctx[b]].sin()
        qr = (x[:Hq] * c + _rotate_half(x[:Hq]) * s_).to(torch.bfloat16).float()
        kk, vv = k1[b, :, :n].float().cpu(), v1[b, :, :n].float().cpu()
        ref = torch.stack([torch.softmax((kk[h // G] @ qr[h]) * HD ** -0.5, 0) @ vv[h // G] for h in range(Hq)])
        _assert_close(outs["2"][b].reshape(Hq, HD), ref, atol=2e-2, rtol=2e-2, what=f"streaming decode attention seq {b}")
    # cross-attention batch (mllama): static keys per sequence, per-sequence key counts, q-norm inside
    Tk = 448
    xk, xv = _randn((B, Hkv, Tk, HD), device, 130), _randn((B, Hkv, Tk, HD), device, 131)
    q = _randn((B, Hq * HD), device, 132)
    qn = _randn((HD,), device, 133)
    nkeys = torch.tensor([int(c) for c in rng.integers(1, Tk, B)], dtype=torch.int32, device=device)
    xs = -(-Tk // hip.DECODE_KEYS_PER_SPLIT)
    res = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("VIS_DECODE_ATTN_STREAM", mode)
        out = torch.empty((B, Hq * HD), dtype=torch.bfloat16, device=device)
        hip.decode_cross_attn_batch(q, qn, xk, xv, nkeys - 1, po, pml, out, Hq, Hkv, HD, xs, HD ** -0.5, 1e-5)
        res[mode] = out.float().cpu()
    assert (res["2"] - res["0"]).abs().max() < 2e-2
    for b in range(min(B, 4)):
        n = int(nkeys[b])
        qq = q[b].float().cpu().reshape(Hq, HD)
        qq = (qq * torch.rsqrt(qq.pow(2).mean(-1, keepdim=True) + 1e-5)).to(torch.bfloat16).float() * qn.float().cpu()
        qq = qq.to(torch.bfloat16).float()
        kk, vv = xk[b, :, :n].float().cpu(), xv[b, :, :n].float().cpu()
        ref = torch.stack([torch.softmax((kk[h // G] @ qq[h]) * HD ** -0.5, 0) @ vv[h // G] for h in range(Hq)])
        _assert_close(res["2"][b].reshape(Hq, HD), ref, atol=2e-2, rtol=2e-2, what=f"streaming cross attention seq {b}")


# ----------------------------------------------------------------------------- K12 sampling
def _sample_many(hip, device, logits, T, n, seed, B=1):
    """n Gumbel-max draws per row through vis_argmax_f32 (the step counter advances by itself: a fresh noise stream per
    draw, exactly as in the decode graph)."""
    V = logits.shape[-1]
    lg = logits.to(device).float().reshape(1, V).repeat(B, 1).contiguous()
    tokens = torch.zeros((B, n), dtype=torch.int32, device=device)
    cur = torch.zeros(B, dtype=torch.int32, device=device)
    step = torch.zeros(B, dtype=torch.int32, device=device)
    wv = torch.empty(256 * B, dtype=torch.float32, device=device)
    wi = torch.empty(256 * B, dtype=torch.int32, device=device)
    for _ in range(n):
        hip.argmax(lg if B > 1 else lg[0], wv, wi, tokens if B > 1 else tokens[0], cur, step, T, seed)
    return tokens.cpu().numpy()


@pytest.mark.parametrize("T", [0.1, 0.2, 1.0])
def test_sampling_follows_softmax_of_logits_over_temperature(hip, device, T):
    """The reference calls its models with temperature 0.1 / 0.2 (utils/config.py:46-49,:66-69; vlm_inspector.py:105-111).
    The Gumbel-max pick must sample softmax(logits / T): chi-square of 20 000 draws over 64 logits against the exact
    probabilities (bins with an expectation below 5 merged), at the reference's operating points and at T = 1."""
    from scipy import stats
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(64, generator=g) * 1.5
    if T < 0.5:     # keep several outcomes alive at low temperature: a cluster of near-equal leaders
        logits[:6] = logits.max() + torch.tensor([0.0, -0.02, -0.05, -0.1, -0.2, -0.3])
    n = 20000
    draws = np.concatenate([_sample_many(hip, device, logits, T, n // 4, seed)[0] for seed in (1, 2, 3, 4)])
    p = torch.softmax(logits.double() / T, dim=0).numpy()
    obs = np.bincount(draws, minlength=64).astype(np.float64)
    exp = p * n
    big = exp >= 5
    o = np.append(obs[big], obs[~big].sum())
    e = np.append(exp[big], exp[~big].sum())
    if e[-1] < 1e-9:
        assert o[-1] == 0
        o, e = o[:-1], e[:-1]
    assert len(e) >= 3, "the test distribution must keep several outcomes"
    chi2, pval = stats.chisquare(o, e * o.sum() / e.sum())
    assert pval > 1e-3, f"T={T}: chi2 {chi2:.1f} over {len(e)} bins, p = {pval:.2e}"


def test_sampling_limits_and_stream_independence(hip, device):
    """T -> 0 reproduces the greedy pick; temperature 0 IS greedy (first index on ties); the slots of a batch draw from
    independent noise streams (seed mixing by slot) while every slot alone follows the same distribution; a fixed seed
    reproduces its draws; no draw ever leaves the support (u = 1 -> +inf noise would: 152 064 logits x 2 000 steps)."""
    g = torch.Generator().manual_seed(9)
    logits = torch.randn(64, generator=g)
    greedy = int(logits.argmax())
    assert (_sample_many(hip, device, logits, 0.0, 8, 5) == greedy).all()
    assert (_sample_many(hip, device, logits, 1e-4, 64, 5) == greedy).all()
    tie = logits.clone(); tie[40] = tie[3] = logits.max() + 1.0
    assert (_sample_many(hip, device, tie, 0.0, 4, 5) == 3).all()
    a = _sample_many(hip, device, logits, 1.0, 512, 11, B=4)
    b = _sample_many(hip, device, logits, 1.0, 512, 11, B=4)
    assert (a == b).all()                                        # reproducible
    for i in range(4):
        for j in range(i + 1, 4):
            agree = float((a[i] == a[j]).mean())                 # independent slots agree with probability sum p^2
            p = torch.softmax(logits.double(), 0)
            assert abs(agree - float((p * p).sum())) < 0.08, (i, j, agree)
    assert not (a[0] == _sample_many(hip, device, logits, 1.0, 512, 12, B=4)[0]).all()      # another seed, other draws
    # full-vocabulary support check: one dominant token at p ~ 1 - 1e-9 must win every draw
    big = torch.full((152064,), -30.0); big[777] = 0.0
    draws = _sample_many(hip, device, big, 1.0, 2000, 3)
    assert (draws == 777).all(), f"{(draws != 777).sum()} draws left the support"


@pytest.mark.parametrize("B", [1, 2, 3, 4])
@pytest.mark.parametrize("N,K,act,fused", [(4608, 3584, 0, "norm+bias"), (3584, 18944, 0, "residual"),
                                           (37888, 3584, 3, "norm"), (1000, 704, 0, "f32")])
def test_gemv_rows_bit_identical_to_single_row(hip, device, B, N, K, act, fused):
    """vis_gemv_bf16_rows / vis_gemv_fp8w_rows: 1..4 input rows share one pass over the weights, and every row equals the
    single-row kernel on it bit for bit (same per-row arithmetic) - so a handful of in-flight sequences decode exactly as
    one would alone.  Shapes: the 7B qkv / down / gate-up projections (down: 4 x 37 KiB of LDS rows) and a ragged one."""
    g = torch.Generator(device="cpu").manual_seed(N + K + B)
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(torch.bfloat16).to(device)
    x = torch.randn((B, K), generator=g).to(torch.bfloat16).to(device)
    n_out = N // 2 if act == 3 else N
    bias = torch.randn((N,), generator=g).to(torch.bfloat16).to(device) if "bias" in fused else None
    nw = (1 + 0.1 * torch.randn((K,), generator=g)).to(torch.bfloat16).to(device) if "norm" in fused else None
    res = torch.randn((B, n_out), generator=g).to(torch.bfloat16).to(device) if "residual" in fused else None
    odt = torch.float32 if fused == "f32" else torch.bfloat16
    for fp8 in (False, True):
        if fp8 and (K % 16 or (act == 3 and N % 64)):
            continue
        wq, sw = hip.quantize_fp8_rows(w) if fp8 else (None, None)
        many = torch.full((B, n_out), 9.0, dtype=odt, device=device)
        if fp8:
            hip.gemv_fp8_rows(x, wq, sw, many, bias=bias, residual=res, norm_w=nw, act=act)
        else:
            hip.gemv_rows(x, w, many, bias=bias, residual=res, norm_w=nw, act=act)
        for b in range(B):
            one = torch.empty((n_out,), dtype=odt, device=device)
            r = res[b] if res is not None else None
            if fp8:
                hip.gemv_fp8(x[b], wq, sw, one, bias=bias, residual=r, norm_w=nw, act=act)
            else:
                hip.gemv(x[b], w, one, bias=bias, residual=r, norm_w=nw, act=act)
            assert torch.equal(many[b], one), f"row {b} of {B} differs from the single-row kernel (fp8={fp8})"


@pytest.mark.parametrize("Hq,Hkv,K,T", [(28, 4, 3584, 4608), (28, 4, 3584, 2560), (2, 1, 256, 1024), (32, 8, 4096, 1024),
                                        (16, 16, 2048, 512), (28, 4, 3584, 6144)])
def test_decode_chain_equals_unchained(hip, device, Hq, Hkv, K, T):
    """vis_decode_chain (qkv projection + RMSNorm + bias -> rope / KV append / split attention + merge -> o projection +
    residual as ONE launch with in-grid hand-offs) against the four launches it replaces (vis_gemv_bf16, vis_decode_attn =
    split + combine, vis_gemv_bf16): y, the merged attention row, the packed projection row and both caches bit for bit,
    at the 7B / 11B / tiny head shapes, at context lengths on and off the 64-key split boundaries up to the last cache row;
    launches repeated back to back on one workspace (granule tags are launch numbers: no reset between launches), status clean.
    T = 4608 is the client's default context (VIS_MAX_CTX: the reference's prompt + its default max_tokens) - a launcher that
    refused that grid (e.g. a register footprint that grew past four waves per SIMD) fails here, not silently in the bench;
    T = 6144 = 96 splits exercises the merge role's path for more than 64 splits and the largest grid the device holds."""
    HD = 128
    nq = (Hq + 2 * Hkv) * HD
    g = torch.Generator(device="cpu").manual_seed(Hq * 1000 + K + T)
    wqkv = (torch.randn((nq, K), generator=g) / K ** 0.5).to(torch.bfloat16).to(device)
    bq = torch.randn((nq,), generator=g).mul(0.1).to(torch.bfloat16).to(device)
    nw = (1 + 0.1 * torch.randn((K,), generator=g)).to(torch.bfloat16).to(device)
    wo = (torch.randn((K, Hq * HD), generator=g) / (Hq * HD) ** 0.5).to(torch.bfloat16).to(device)
    kc0, vc0 = _randn((Hkv, T, HD), device, 301), _randn((Hkv, T, HD), device, 302)
    ang = torch.rand((T, HD // 2), generator=g) * 6.28
    emb = torch.cat((ang, ang), -1)
    cos_t, sin_t = emb.cos().to(device).contiguous(), emb.sin().to(device).contiguous()
    ns = -(-T // hip.DECODE_KEYS_PER_SPLIT)
    po = torch.empty(Hq * ns * HD, dtype=torch.float32, device=device)
    pml = torch.empty(Hq * ns * 2, dtype=torch.float32, device=device)
    ws, sync = hip.decode_chain_state(device, Hq, Hkv, ns)
    assert hip.decode_chain_supported(Hq, Hkv, HD, K)
    launches = 0
    ctxs = (0, 1, 62, 63, 64, 65, 127, 128, 700, T // 2 + 3, T - 2, T - 1) if T != 6144 else (63, 4095, 4097, 5000, T - 1)
    for ctx in sorted({c for c in ctxs if c < T}):
        x = _randn((K,), device, 400 + ctx)
        step = torch.tensor([ctx], dtype=torch.int32, device=device)
        # ---- the four launches
        k1, v1 = kc0.clone(), vc0.clone()
        qkv1 = torch.empty(nq, dtype=torch.bfloat16, device=device)
        att1 = torch.empty(Hq * HD, dtype=torch.bfloat16, device=device)
        y1 = torch.empty(K, dtype=torch.bfloat16, device=device)
        hip.gemv(x, wqkv, qkv1, bias=bq, norm_w=nw, eps=1e-6)
        hip.decode_attn(qkv1, cos_t, sin_t, k1, v1, step, po, pml, att1, Hq, Hkv, HD, ns, HD ** -0.5)
        hip.gemv(att1, wo, y1, residual=x)
        # ---- one launch (twice: the second must not depend on anything the first left behind)
        for rep in range(2):
            k2, v2 = kc0.clone(), vc0.clone()
            y2 = torch.full((K,), 3.0, dtype=torch.bfloat16, device=device)
            if rep == 0:
                hip.decode_chain(x, wqkv, bq, nw, wo, y2, cos_t, sin_t, k2, v2, step, ws, sync, Hq, Hkv, HD, ns, HD ** -0.5, 1e-6)
            else:       # the layer input as a row of a table picked by a device index (the first layer: embedding lookup inside)
                table = torch.stack([_randn((K,), device, 77), x, _randn((K,), device, 78)])
                hip.decode_chain(table, wqkv, bq, nw, wo, y2, cos_t, sin_t, k2, v2, step, ws, sync, Hq, Hkv, HD, ns, HD ** -0.5,
                                 1e-6, x_index=torch.tensor([1], dtype=torch.int32, device=device))
            launches += 1
            torch.cuda.synchronize()
            qkv2, att2 = hip.decode_chain_rows(ws, Hq, Hkv)
            assert int(sync[hip.CHAIN_STATUS_WORD]) == 0, f"ctx {ctx}: a wait inside the chained launch timed out"
            assert int(sync[0]) == launches, "the launch counter of the sync block must advance once per launch"
            assert torch.equal(qkv2, qkv1), f"ctx {ctx}: projection row differs"
            assert torch.equal(k2, k1) and torch.equal(v2, v1), f"ctx {ctx}: KV append differs"
            assert torch.equal(att2, att1), f"ctx {ctx}: merged attention differs " \
                                            f"(max {float((att2.float() - att1.float()).abs().max())})"
            assert torch.equal(y2, y1), f"ctx {ctx}: o projection differs"
    # fp32 reference of the whole head at the last context (the unchained kernels are tested against it elsewhere too)
    xf = x.float().cpu()
    xn = (xf * torch.rsqrt(xf.pow(2).mean() + 1e-6)).to(torch.bfloat16).float() * nw.float().cpu()
    qkv_ref = (wqkv.float().cpu() @ xn + bq.float().cpu())
    _assert_close(qkv2, qkv_ref, atol=3e-2, rtol=2e-2, what="chained qkv projection vs fp32")


def test_decode_chain_refuses_unsupported_shapes(hip, device):
    """Shapes outside the chained form are refused by the launcher before any launch (the engine then issues four launches)."""
    assert not hip.decode_chain_supported(28, 4, 64, 3584)       # head_dim
    assert not hip.decode_chain_supported(64, 8, 128, 8192)      # hidden above one K-segment per row
    assert not hip.decode_chain_supported(28, 3, 128, 3584)      # heads not a multiple of kv heads
    # a supported head shape whose grid (projection + Hkv x splits + merge workgroups) exceeds what the device holds resident:
    # Llama-3.2-11B text shape at a 16 k context = 768 + 8 * 256 + 32 workgroups
    Hq, Hkv, K, T, HD = 32, 8, 4096, 16384, 128
    ns = T // hip.DECODE_KEYS_PER_SPLIT
    nq = (Hq + 2 * Hkv) * HD
    z = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device=device)
    ws, sync = hip.decode_chain_state(device, Hq, Hkv, ns)
    tab = torch.zeros((T, HD), dtype=torch.float32, device=device)
    with pytest.raises(hip.ChainRefused):      # VIS_ERR_UNSUPPORTED, the one status the engine answers with the four launches
        hip.decode_chain(z(K), z(nq, K), None, z(K), z(K, Hq * HD), z(K), tab, tab, z(Hkv, T, HD), z(Hkv, T, HD),
                         torch.zeros(1, dtype=torch.int32, device=device), ws, sync, Hq, Hkv, HD, ns, HD ** -0.5, 1e-6)
    torch.cuda.synchronize()
    assert int(sync[0]) == 0 and int(sync[hip.CHAIN_STATUS_WORD]) == 0        # nothing was launched
    # a caller bug is NOT a refusal: a misaligned pointer must surface as an error, never as a silent fall-back
    Hq, Hkv, K, T = 28, 4, 3584, 256
    ns, nq = T // hip.DECODE_KEYS_PER_SPLIT, (Hq + 2 * Hkv) * HD
    ws, sync = hip.decode_chain_state(device, Hq, Hkv, ns)
    tab = torch.zeros((T, HD), dtype=torch.float32, device=device)
    xbad = torch.zeros(K + 1, dtype=torch.bfloat16, device=device)[1:]     # 2-byte aligned
    with pytest.raises(hip.HipLibraryError) as ei:
        hip.decode_chain(xbad, z(nq, K), None, z(K), z(K, Hq * HD), z(K), tab, tab, z(Hkv, T, HD), z(Hkv, T, HD),
                         torch.zeros(1, dtype=torch.int32, device=device), ws, sync, Hq, Hkv, HD, ns, HD ** -0.5, 1e-6)
    assert not isinstance(ei.value, hip.ChainRefused)


def _chain_case(device, Hq, Hkv, K, T, seed):
    HD = 128
    nq = (Hq + 2 * Hkv) * HD
    g = torch.Generator(device="cpu").manual_seed(seed)
    c = {"wqkv": (torch.randn((nq, K), generator=g) / K ** 0.5).to(torch.bfloat16).to(device),
         "bq": torch.randn((nq,), generator=g).mul(0.1).to(torch.bfloat16).to(device),
         "nw": (1 + 0.1 * torch.randn((K,), generator=g)).to(torch.bfloat16).to(device),
         "wo": (torch.randn((K, Hq * HD), generator=g) / (Hq * HD) ** 0.5).to(torch.bfloat16).to(device),
         "kc": _randn((Hkv, T, HD), device, seed + 1), "vc": _randn((Hkv, T, HD), device, seed + 2)}
    ang = torch.rand((T, HD // 2), generator=g) * 6.28
    emb = torch.cat((ang, ang), -1)
    c["cos"], c["sin"] = emb.cos().to(device).contiguous(), emb.sin().to(device).contiguous()
    return c


def test_decode_chain_epoch_wrap(hip, device):
    """The granule tag of a chained launch is the sync block's 32-bit launch counter + 1; tag 0 is what a never-written
    (zero-initialised) granule carries.  Six launches across the wrap (counter seeded at 0xFFFFFFFD: tags ...FE, ...FF, then 1
    instead of 0, 2, 3, 4) on a FRESH workspace - a launch that ran with tag 0 would accept unwritten granules at once - each
    bit-identical to the four launches."""
    Hq, Hkv, K, T, HD = 28, 4, 3584, 1024, 128
    nq = (Hq + 2 * Hkv) * HD
    c = _chain_case(device, Hq, Hkv, K, T, 900)
    ns = -(-T // hip.DECODE_KEYS_PER_SPLIT)
    po = torch.empty(Hq * ns * HD, dtype=torch.float32, device=device)
    pml = torch.empty(Hq * ns * 2, dtype=torch.float32, device=device)
    ws, sync = hip.decode_chain_state(device, Hq, Hkv, ns)
    sync[0] = -3                                                           # 0xFFFFFFFD
    expect = [-2, -1, 1, 2, 3, 4]                                          # the counter after each launch (int32 view)
    for i, ctx in enumerate((5, 130, 700, 64, 1000, 333)):
        x = _randn((K,), device, 950 + i)
        step = torch.tensor([ctx], dtype=torch.int32, device=device)
        k1, v1, k2, v2 = c["kc"].clone(), c["vc"].clone(), c["kc"].clone(), c["vc"].clone()
        qkv1 = torch.empty(nq, dtype=torch.bfloat16, device=device)
        att1 = torch.empty(Hq * HD, dtype=torch.bfloat16, device=device)
        y1 = torch.empty(K, dtype=torch.bfloat16, device=device)
        y2 = torch.full((K,), 3.0, dtype=torch.bfloat16, device=device)
        hip.gemv(x, c["wqkv"], qkv1, bias=c["bq"], norm_w=c["nw"], eps=1e-6)
        hip.decode_attn(qkv1, c["cos"], c["sin"], k1, v1, step, po, pml, att1, Hq, Hkv, HD, ns, HD ** -0.5)
        hip.gemv(att1, c["wo"], y1, residual=x)
        hip.decode_chain(x, c["wqkv"], c["bq"], c["nw"], c["wo"], y2, c["cos"], c["sin"], k2, v2, step, ws, sync, Hq, Hkv, HD, ns,
                         HD ** -0.5, 1e-6)
        torch.cuda.synchronize()
        assert int(sync[hip.CHAIN_STATUS_WORD]) == 0, f"launch {i}: a wait timed out"
        assert int(sync[0]) == expect[i], f"launch {i}: counter {int(sync[0])}, expected {expect[i]}"
        assert torch.equal(y2, y1) and torch.equal(k2, k1) and torch.equal(v2, v1), f"launch {i} (tag across the wrap) differs"


def test_decode_chain_fails_fast_after_a_stall(hip, device):
    """Once the status word of a sync block is raised (a bounded wait of an earlier launch gave up), every further chained launch
    on that block returns at once: no waits, nothing computed, the launch counter still advances (a
    stranded request used to spin through every remaining launch's bounds - minutes - before the host looked at the word)."""
    import time
    Hq, Hkv, K, T, HD = 28, 4, 3584, 4608, 128
    c = _chain_case(device, Hq, Hkv, K, T, 990)
    ns = -(-T // hip.DECODE_KEYS_PER_SPLIT)
    ws, sync = hip.decode_chain_state(device, Hq, Hkv, ns)
    x = _randn((K,), device, 991)
    step = torch.tensor([2300], dtype=torch.int32, device=device)
    y = torch.full((K,), 3.0, dtype=torch.bfloat16, device=device)
    args = (x, c["wqkv"], c["bq"], c["nw"], c["wo"], y, c["cos"], c["sin"], c["kc"], c["vc"], step, ws, sync, Hq, Hkv, HD, ns,
            HD ** -0.5, 1e-6)
    hip.decode_chain(*args)                                               # a healthy launch first
    torch.cuda.synchronize()
    assert int(sync[hip.CHAIN_STATUS_WORD]) == 0 and int(sync[0]) == 1 and not torch.equal(y, torch.full_like(y, 3.0))
    sync[hip.CHAIN_STATUS_WORD] = 1                                       # "a wait gave up"
    y.fill_(3.0)
    kc0 = c["kc"].clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 28 * 128                                                          # the launches of 128 further tokens
    for _ in range(n):
        hip.decode_chain(*args)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(sync[0]) == 1 + n and int(sync[hip.CHAIN_STATUS_WORD]) == 1
    assert torch.equal(y, torch.full_like(y, 3.0)) and torch.equal(c["kc"], kc0), "a launch behind a raised status word computed"
    # 3584 launches of ~860 workgroups that only read one word: host-launch bound (~5-10 us each); the r04 kernel spun
    # up to three bounds of 65 536 polls in every one of them
    assert dt < 0.5, f"{n} launches behind a raised status word took {dt:.3f} s"


@pytest.mark.parametrize("N,K,temperature", [(152064, 3584, 0.0), (152064, 3584, 0.7), (512, 256, 0.0), (1001, 704, 0.3)])
def test_gemv_argmax_equals_two_stage(hip, device, N, K, temperature):
    """vis_gemv_bf16_argmax (lm_head with the pick's first stage in its epilogue + the merging launch) against vis_gemv_bf16 +
    vis_argmax_f32: the same f32 logits bit for bit, the same pick (greedy with ties, and Gumbel-max sampled with the same
    seed / step), the same device-side bookkeeping (tokens[step], cur_token, step + 1)."""
    g = torch.Generator(device="cpu").manual_seed(N + K)
    w = (torch.randn((N, K), generator=g) / K ** 0.5).to(torch.bfloat16).to(device)
    if temperature == 0.0:
        w[N // 3] = w[N // 2]                    # an exact tie: the lower index must win in both forms
    nw = (1 + 0.1 * torch.randn((K,), generator=g)).to(torch.bfloat16).to(device)
    for trial in range(3):
        x = _randn((K,), device, 900 + trial)
        if temperature == 0.0 and trial == 0:
            x = (w[N // 2].float() * 4).to(torch.bfloat16)          # make the tied rows the maximum
        out = {}
        for fused in (False, True):
            logits = torch.zeros(N, dtype=torch.float32, device=device)
            wv = torch.empty(2048, dtype=torch.float32, device=device)
            wi = torch.empty(2048, dtype=torch.int32, device=device)
            tokens = torch.zeros(64, dtype=torch.int32, device=device)
            cur = torch.zeros(1, dtype=torch.int32, device=device)
            step = torch.tensor([5 + trial], dtype=torch.int32, device=device)
            if fused:
                hip.gemv_argmax(x, w, logits, wv, wi, tokens, cur, step, norm_w=nw, temperature=temperature, seed=1234)
            else:
                hip.gemv(x, w, logits, norm_w=nw)
                hip.argmax(logits, wv, wi, tokens, cur, step, temperature, 1234)
            out[fused] = (logits.clone(), tokens.clone(), int(cur), int(step))
        assert torch.equal(out[True][0], out[False][0]), "logits differ"
        assert out[True][2] == out[False][2] and out[True][3] == out[False][3] == 6 + trial
        assert torch.equal(out[True][1], out[False][1])
        if temperature == 0.0:
            assert out[True][2] == int(out[True][0].argmax())        # torch.argmax: first index on ties
            if trial == 0:
                assert out[True][2] == N // 3


@pytest.mark.parametrize("Hq,Hkv,B,T,P", [(28, 4, 32, 2560, 960), (32, 8, 16, 1024, 704 // 64 * 64), (28, 4, 64, 1536, 64)])
def test_decode_attn_shared_prefix_is_bit_identical(hip, device, Hq, Hkv, B, T, P):
    """vis_decode_attn_shared: the first P cached keys of every sequence are copies of one text prefix (what a batch inspection's
    prompt passes leave in the slots); reading them from sequence 0's cache must change nothing - outputs and appended rows
    equal to vis_decode_attn bit for bit, ragged contexts, and the result must really come from sequence 0's copy (poisoning
    the OTHER sequences' prefix rows does not change it)."""
    HD = 128
    rng = np.random.default_rng(9)
    ctx = [int(c) for c in rng.integers(P, T - 1, B)]
    ctx[0], ctx[1] = P, T - 2
    kc = _randn((B, Hkv, T, HD), device, 520)
    vc = _randn((B, Hkv, T, HD), device, 521)
    kc[:, :, :P] = kc[0:1, :, :P]
    vc[:, :, :P] = vc[0:1, :, :P]
    g = torch.Generator().manual_seed(522)
    ang = torch.rand((T, HD // 2), generator=g) * 6.28
    emb = torch.cat((ang, ang), -1)
    cos_t = emb.cos().to(device).unsqueeze(0).expand(B, -1, -1)
    sin_t = emb.sin().to(device).unsqueeze(0).expand(B, -1, -1)
    qkv = _randn((B, (Hq + 2 * Hkv) * HD), device, 523)
    step = torch.tensor(ctx, dtype=torch.int32, device=device)
    ns = -(-T // hip.DECODE_KEYS_PER_SPLIT)
    po = torch.empty(B * Hq * ns * HD, dtype=torch.float32, device=device)
    pml = torch.empty(B * Hq * ns * 2, dtype=torch.float32, device=device)
    assert Hkv * B >= 128                       # the streaming form (the only one that reads the shared copy)
    res = {}
    for name, shared, poison in (("own", 0, False), ("shared", P, False), ("poisoned", P, True)):
        k1, v1 = kc.clone(), vc.clone()
        if poison:
            k1[1:, :, :P] = 77.0
            v1[1:, :, :P] = -55.0
        out = torch.full((B, Hq * HD), 7.0, dtype=torch.bfloat16, device=device)
        hip.decode_attn(qkv, cos_t, sin_t, k1, v1, step, po, pml, out, Hq, Hkv, HD, ns, HD ** -0.5, shared_len=shared)
        res[name] = (out.clone(), k1[:, :, P:].clone(), v1[:, :, P:].clone())
    for name in ("shared", "poisoned"):
        assert torch.equal(res[name][0], res["own"][0]), f"{name}: attention output differs"
        assert torch.equal(res[name][1], res["own"][1]) and torch.equal(res[name][2], res["own"][2]), f"{name}: KV append differs"
    with pytest.raises(hip.HipLibraryError):
        hip.decode_attn(qkv, cos_t, sin_t, kc, vc, step, po, pml, out, Hq, Hkv, HD, ns, HD ** -0.5, shared_len=100)   # not x 64


@pytest.mark.gpu
@pytest.mark.parametrize("Hq,Hkv,B,fp8,bias,P", [(28, 4, 64, False, True, 192), (28, 4, 40, False, True, 0), (28, 4, 4, False, True, 0),
                                                (28, 4, 4, True, True, 0), (28, 4, 64, True, True, 128), (32, 8, 17, False, False, 0),
                                                (32, 8, 32, True, False, 64), (32, 8, 8, False, False, 0)])
def test_decode_attn_parts_equals_finalise_then_attention(hip, device, Hq, Hkv, B, fp8, bias, P):
    """vis_decode_attn_parts (r05): the attention launch finalises the qkv columns it reads from the batched projection's partial
    slabs.  Outputs and appended K / V rows must equal skinny_finalize[_fp8] + decode_attn bit for bit - split form (small batch)
    and streaming form, 16- / 32- / 64-row slabs, bf16 and raw-e4m3 partials, with and without bias and shared prefix."""
    HD, K, T = 128, 1024, 448
    nq = (Hq + 2 * Hkv) * HD
    x = _randn((B, K), device, 710, 1.5)
    w = _randn((nq, K), device, 711, 1.0 / math.sqrt(K))
    b = _randn((nq,), device, 712) if bias else None
    part = torch.full((16 * hip.part_rows(B) * nq,), float("nan"), dtype=torch.float32, device=device)
    if fp8:
        xq, sx = hip.quant_rows_fp8(x)
        wq, sw = hip.quantize_fp8_rows(w)
        ks = hip.decode_gemm_fp8(xq, sx, wq, sw, part=part)
    else:
        sx = sw = None
        ks = hip.decode_gemm(x, w, part=part)
    rng = np.random.default_rng(713)
    ctx = [int(c) for c in rng.integers(max(P, 1), T - 1, B)]
    ctx[0], ctx[-1] = max(P, 1), T - 2
    kc = _randn((B, Hkv, T, HD), device, 714)
    vc = _randn((B, Hkv, T, HD), device, 715)
    if P:
        kc[:, :, :P] = kc[0:1, :, :P]
        vc[:, :, :P] = vc[0:1, :, :P]
    g = torch.Generator().manual_seed(716)
    ang = torch.rand((T, HD // 2), generator=g) * 6.28
    emb = torch.cat((ang, ang), -1)
    cos_t = emb.cos().to(device).unsqueeze(0).expand(B, -1, -1)
    sin_t = emb.sin().to(device).unsqueeze(0).expand(B, -1, -1)
    step = torch.tensor(ctx, dtype=torch.int32, device=device)
    ns = -(-T // hip.DECODE_KEYS_PER_SPLIT)
    po = torch.empty(B * Hq * ns * HD, dtype=torch.float32, device=device)
    pml = torch.empty(B * Hq * ns * 2, dtype=torch.float32, device=device)
    # the pair
    qkv = torch.empty((B, nq), dtype=torch.bfloat16, device=device)
    if fp8:
        hip.skinny_finalize_fp8(part, ks, qkv, nq, sx=sx, sw=sw, bias=b)
    else:
        hip.skinny_finalize(part, ks, qkv, nq, bias=b)
    k1, v1 = kc.clone(), vc.clone()
    out1 = torch.full((B, Hq * HD), 7.0, dtype=torch.bfloat16, device=device)
    hip.decode_attn(qkv, cos_t, sin_t, k1, v1, step, po, pml, out1, Hq, Hkv, HD, ns, HD ** -0.5, shared_len=P)
    # one launch
    k2, v2 = kc.clone(), vc.clone()
    out2 = torch.full((B, Hq * HD), -3.0, dtype=torch.bfloat16, device=device)
    hip.decode_attn_parts(part, ks, cos_t, sin_t, k2, v2, step, po, pml, out2, Hq, Hkv, HD, ns, HD ** -0.5, bias=b, sx=sx, sw=sw,
                          shared_len=P)
    assert torch.isfinite(out1.float()).all()
    assert torch.equal(out1, out2), f"max diff {(out1.float() - out2.float()).abs().max().item()}"
    assert torch.equal(k1, k2) and torch.equal(v1, v2), "appended K / V rows differ"
    with pytest.raises(hip.HipLibraryError):      # scales come in pairs
        hip.decode_attn_parts(part, ks, cos_t, sin_t, k2, v2, step, po, pml, out2, Hq, Hkv, HD, ns, HD ** -0.5,
                              sx=torch.ones(B, dtype=torch.float32, device=device))


@pytest.mark.gpu
@pytest.mark.parametrize("k,Hq,Hkv,S,P", [(4, 28, 4, 1289, 960), (3, 8, 2, 200, 64), (8, 4, 4, 77, 0), (1, 28, 4, 300, 128)])
def test_group_rope_split_and_pair_attention_equal_per_request_launches(hip, device, k, Hq, Hkv, S, P):
    """vis_qkv_rope_split_many / vis_attn_prefill_pairs_many (r05): the requests of a stacked prompt-pass group as ONE launch each.
    Everything a request's launches would have written - q, its cache slot's K / V rows, its V^T columns, its attention rows - must
    be bit-identical, arbitrary (non-consecutive) cache slots, nothing else touched."""
    HD, L, T = 128, 2, P + S + 40
    li = 1
    ld = (T + 63) // 64 * 64
    nq = (Hq + 2 * Hkv) * HD
    slots_total = k + 3
    slots = torch.randperm(slots_total, generator=torch.Generator().manual_seed(729 + k))[:k].tolist()   # arbitrary, distinct
    qkv = _randn((k * S, nq), device, 730)
    g = torch.Generator().manual_seed(731)
    ang = torch.rand((S, HD // 2), generator=g) * 6.28
    emb = torch.cat((ang, ang), -1)
    cos, sin = emb.cos().to(device).contiguous(), emb.sin().to(device).contiguous()
    kc0 = _randn((slots_total, L, Hkv, T, HD), device, 732)
    vc0 = _randn((slots_total, L, Hkv, T, HD), device, 733)
    vt0 = _randn((k, L, Hkv, HD, ld), device, 734)
    work = hip.make_attn_pairs(P, P + S, device)
    scale = HD ** -0.5
    # per request
    kc1, vc1, vt1 = kc0.clone(), vc0.clone(), vt0.clone()
    q1 = torch.zeros((k, Hq, S, HD), dtype=torch.bfloat16, device=device)
    o1 = torch.zeros((k * S, Hq * HD), dtype=torch.bfloat16, device=device)
    for j, sl in enumerate(slots):
        rows = slice(j * S, (j + 1) * S)
        hip.qkv_rope_split(qkv[rows], cos, sin, q1[j], kc1[sl][li], vc1[sl][li], vt1[j][li], Hq, Hkv, HD, k_pos0=P, vt_col0=P // 64 * 64)
        hip.attn_prefill_pairs(q1[j], kc1[sl][li], vt1[j][li], o1[rows], work, scale, q_row0=P)
    # one launch each
    kc2, vc2, vt2 = kc0.clone(), vc0.clone(), vt0.clone()
    q2 = torch.zeros_like(q1)
    o2 = torch.zeros_like(o1)
    kv_off = [sl * kc2.stride(0) + li * kc2.stride(1) for sl in slots]
    hip.qkv_rope_split_many(qkv, cos, sin, q2, kc2, vc2, vt2[:, li], Hq, Hkv, HD, kv_off, T, k_pos0=P, vt_col0=P // 64 * 64)
    hip.attn_prefill_pairs_many(q2, kc2, vt2[:, li], o2, work, scale, kv_off, T, q_row0=P)
    assert torch.equal(q1, q2), "q differs"
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2), "cache rows differ (or a foreign slot / layer was touched)"
    assert torch.equal(vt1, vt2), "V^T differs"
    assert torch.isfinite(o1.float()).all() and torch.equal(o1, o2), "attention rows differ"
    with pytest.raises(hip.HipLibraryError):
        hip.attn_prefill_pairs_many(q2, kc2, vt2[:, li], o2, work, scale, [o + 4 for o in kv_off], T, q_row0=P)   # offsets are x 8
    with pytest.raises(hip.HipLibraryError):
        hip.qkv_rope_split_many(qkv, cos, sin, q2, kc2, vc2, vt2[:, li], Hq, Hkv, HD, [kc2.numel()] * k, T, k_pos0=P)   # outside the cache


@pytest.mark.gpu
@pytest.mark.parametrize("tokens,heads,extra", [(6432, 8, 8), (37, 4, 0), (5, 6, 2), (1, 16, 16)])
def test_rmsnorm_heads_equals_rmsnorm_per_head(hip, device, tokens, heads, extra):
    """vis_rmsnorm_heads_bf16 (r05): mllama's k_norm over the key heads of a packed [k | v] projection row in one launch - every
    (token, head) slice bit-identical to vis_rmsnorm_bf16 on the strided slice, the columns behind the heads untouched."""
    D = 128
    x = _randn((tokens, (heads + extra) * D), device, 740, 1.7)
    w = _randn((D,), device, 741)
    ref = x.clone()
    for h in range(heads):
        hip.rmsnorm(x[:, h * D:(h + 1) * D], w, 1e-5, out=ref[:, h * D:(h + 1) * D])
    got = x.clone()
    hip.rmsnorm_heads(got, w, heads, 1e-5)
    assert torch.equal(got, ref)
    if extra:
        assert torch.equal(got[:, heads * D:], x[:, heads * D:])
    with pytest.raises(hip.HipLibraryError):
        hip.rmsnorm_heads(got, w, heads + extra + 1, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("k,Hq,Hkv,S,causal", [(4, 32, 8, 300, False), (3, 8, 2, 200, True), (8, 4, 4, 77, False)])
def test_group_attention_rows_equal_per_request_launches(hip, device, k, Hq, Hkv, S, causal):
    """vis_attn_prefill_rows_many (r05): the generic head_dim-128 kernel over the requests of a group, one work list PER request
    (different key counts - the Auditor's cross-attention over 1..4 tiles), arbitrary cache blocks: rows bit-identical."""
    HD, L, T = 128, 3, 1700
    li = 2
    ld = (T + 63) // 64 * 64
    slots_total = k + 2
    slots = torch.randperm(slots_total, generator=torch.Generator().manual_seed(750 + k))[:k].tolist()
    q = _randn((k, Hq, S, HD), device, 751)
    kc = _randn((slots_total, L, Hkv, T, HD), device, 752)
    vt = _randn((k, Hkv, HD, ld), device, 753)
    nkeys = [max(S, T - 300 * j) for j in range(k)]
    items = [[(q0, min(128, S - q0), 0, nkeys[j]) for q0 in range(0, S, 128)] for j in range(k)]
    work = torch.tensor(items, dtype=torch.int32, device=device).reshape(k, -1, 4).contiguous()
    scale = HD ** -0.5
    o1 = torch.zeros((k * S, Hq * HD), dtype=torch.bfloat16, device=device)
    for j, sl in enumerate(slots):
        hip.attn_prefill(q[j], kc[sl][li], vt[j], o1[j * S:(j + 1) * S], work[j], causal, scale)
    o2 = torch.zeros_like(o1)
    kv_off = [sl * kc.stride(0) + li * kc.stride(1) for sl in slots]
    hip.attn_prefill_many(q, kc, vt, o2, work, causal, scale, kv_off, T)
    assert torch.isfinite(o1.float()).all() and torch.equal(o1, o2)
