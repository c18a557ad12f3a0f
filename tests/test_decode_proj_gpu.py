"""vis_decode_proj_* (csrc/decode_stream.hip: stream-K ranges + last-arriver reduction) and vis_decode_proj_colpar_*
(csrc/decode_colpar.hip: whole-K column slabs, nothing reduced across workgroups): the batched-decode projection with its
row-wise epilogue in one launch, against fp32 PyTorch statements of the same arithmetic; bitwise repeatability and row
independence; the fp8 forms on MX blocks against oracle/mx_ref.py."""
import pytest
import torch

from oracle import mx_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from vision_inspection_system_amd import hip as h
    h.load()
    return h


def _rt(x):   # round through bf16
    return x.to(torch.bfloat16).float()


def _mk(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


def _ref_epilogue(acc, mode, rs, bias, res, nw):
    """acc [B, N] f32 = x w^T -> (main, weighted, ssq tiles) as the kernel defines them."""
    B, N = acc.shape
    v = acc * rs[:, None]
    if mode == 0:
        if bias is not None:
            v = v + bias.float()[None, :]
        return v, None, None
    if mode == 1:
        g = v.view(B, N // 32, 2, 16)[:, :, 0, :].reshape(B, N // 2)
        u = v.view(B, N // 32, 2, 16)[:, :, 1, :].reshape(B, N // 2)
        return _rt(torch.nn.functional.silu(g) * u), None, None
    y = _rt(acc + res.float())
    yw = _rt(y * nw.float()[None, :])
    units = (N + 31) // 32
    pad = torch.zeros((B, units * 32), dtype=torch.float32)
    pad[:, :N] = y * y
    return y, yw, pad.view(B, units, 32).sum(-1).T.contiguous()     # [units, B]


SHAPES = [  # (N, K, mode, what)
    (4608, 3584, 0, "7B qkv (bias, rs)"), (3584, 3584, 2, "7B o"), (37888, 3584, 1, "7B gate/up"),
    (3584, 18944, 2, "7B down"), (152064, 3584, 0, "7B lm_head f32"), (512, 256, 0, "tiny qkv"), (256, 704, 2, "tiny down"),
    (1408, 256, 1, "tiny gate/up"), (1000, 192, 0, "ragged N"), (384000, 64, 0, "12 tiles per workgroup: sets overflow"),
    (6144, 128, 2, "short tiles, many per range"),
]


def _covered(hip, form, N, mode, mx):
    return form == "streamk" or bool(hip.load().vis_decode_proj_colpar_covers(N, mode, 1 if mx else 0))


@pytest.mark.parametrize("form", ["streamk", "colpar"])
@pytest.mark.parametrize("N,K,mode,what", SHAPES)
@pytest.mark.parametrize("B", [1, 4, 16, 17, 33, 64])
def test_decode_proj_bf16_matches_fp32(hip, device, N, K, mode, what, B, form):
    if N > 100000 and B not in (4, 64):
        pytest.skip("large-N shapes at two batch sizes only")
    if not _covered(hip, form, N, mode, False):
        pytest.skip("shape outside the column-parallel form (N % 32, or more than five units per workgroup)")
    seed = N + K + B
    x, w = _mk((B, K), seed, 1.0), _mk((N, K), seed + 1, K ** -0.5)
    f32_out = what.endswith("f32")
    bias = _mk((N,), seed + 2, 0.5) if mode == 0 and not f32_out else None
    res = _mk((B, N), seed + 3) if mode == 2 else None
    nw = (1 + 0.1 * _mk((N,), seed + 4).float()).to(torch.bfloat16) if mode == 2 else None
    use_rs = mode != 2
    tiles_in = min(128, (K + 31) // 32)
    ssq_in = (torch.rand((tiles_in, 64), generator=torch.Generator().manual_seed(seed + 5)) * 50 + 5) if use_rs else None
    rs = torch.rsqrt(ssq_in[:, :B].sum(0) / K + 1e-6) if use_rs else torch.ones(B)
    xd, wd = x.to(device), w.to(device)
    acc = (xd.float() @ wd.float().T).cpu()        # fp32 product of the bf16 operands (reference arithmetic, on the device for speed)
    main, weighted, ssq = _ref_epilogue(acc, mode, rs, bias, res, nw)
    n_out = N // 2 if mode == 1 else N
    ws = hip.decode_proj_ws(device, B, N, K)
    out = torch.full((B, n_out), 7.0, dtype=torch.float32 if f32_out else torch.bfloat16, device=device)
    out_w = torch.full((B, n_out), 7.0, dtype=torch.bfloat16, device=device) if mode == 2 else None
    ssq_out = torch.full(((N + 31) // 32, 64), -1.0, dtype=torch.float32, device=device) if mode == 2 else None
    kw = dict(out=out, out_w=out_w, bias=bias.to(device) if bias is not None else None,
              residual=res.to(device) if res is not None else None, norm_w=nw.to(device) if nw is not None else None,
              ssq_in=ssq_in.to(device) if use_rs else None, ssq_out=ssq_out, norm_dim=K if use_rs else 0, eps=1e-6, form=form)
    hip.decode_proj(xd, wd, ws, mode, **kw)
    torch.cuda.synchronize()
    assert int(ws[:16384].view(torch.int32).abs().sum()) == 0, "arrival counters must be back at zero after the launch"
    got = out.float().cpu()
    tol = dict(atol=2e-2, rtol=2e-2) if not f32_out else dict(atol=2e-3, rtol=2e-3)
    assert torch.allclose(got, main if f32_out else _rt(main), **tol), \
        f"{what} B={B}: max err {float((got - main).abs().max())}"
    if mode == 2:
        assert torch.allclose(out_w.float().cpu(), weighted, atol=3e-2, rtol=2e-2)
        # the weighted copy is EXACTLY bf16(y * nw) of the y the kernel wrote
        assert torch.equal(out_w.float().cpu(), _rt(got * nw.float()[None, :]))
        s_ref = torch.zeros_like(ssq)
        yy = torch.zeros((B, ((N + 31) // 32) * 32))
        yy[:, :N] = got * got
        s_ref = yy.view(B, -1, 32).sum(-1).T
        assert torch.allclose(ssq_out[:, :B].cpu(), s_ref, rtol=1e-5, atol=1e-6), "tile sums of squares of the written y"
        assert bool((ssq_out[:, B:] == -1.0).all()), "rows >= B of the ssq buffer must not be touched"
    # bitwise repeatable on the same workspace, and a row's result does not depend on the batch it sits in
    first = out.clone()
    hip.decode_proj(xd, wd, ws, mode, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out, first)
    if B > 1:
        for Bs, r in ((1, B - 1), (min(B, 16), 0)):
            o1 = torch.empty((Bs, n_out), dtype=out.dtype, device=device)
            kw1 = dict(kw, out=o1, out_w=torch.empty_like(out_w[:Bs]) if out_w is not None else None,
                       residual=kw["residual"][r:r + Bs] if res is not None else None,
                       ssq_in=None, norm_dim=0,
                       ssq_out=torch.empty_like(ssq_out) if ssq_out is not None else None)
            if use_rs:   # the row's own ssq column moved to column 0..
                s1 = torch.zeros((tiles_in, 64), dtype=torch.float32, device=device)
                s1[:, :Bs] = kw["ssq_in"][:, r:r + Bs]
                kw1.update(ssq_in=s1, norm_dim=K)
            hip.decode_proj(xd[r:r + Bs], wd, ws, mode, **kw1)
            torch.cuda.synchronize()
            assert torch.equal(o1, first[r:r + Bs]), f"{what}: rows {r}..{r + Bs} differ between batch {B} and batch {Bs}"


def test_decode_prep_rows(hip, device):
    H, V, B = 3584, 1000, 37
    table, nw = _mk((V, H), 1).to(device), (1 + 0.1 * _mk((H,), 2).float()).to(torch.bfloat16).to(device)
    ids = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)
    ids[0], ids[1] = -5, V + 9                      # clamped like vis_gather_rows
    x = torch.empty((B, H), dtype=torch.bfloat16, device=device)
    xw, ssq = torch.empty_like(x), torch.full((H // 32, 64), -1.0, dtype=torch.float32, device=device)
    xq = torch.zeros((B, H), dtype=torch.uint8, device=device)
    xqs = torch.zeros((B, H // 32), dtype=torch.uint8, device=device)
    hip.decode_prep_rows(table, ids.to(device), nw, x, xw, ssq, xq, xqs)
    rows = table[ids.clamp(0, V - 1).long().to(device)]
    assert torch.equal(x, rows)
    w_ref = _rt(rows.float().cpu() * nw.float().cpu()[None, :])
    assert torch.equal(xw.float().cpu(), w_ref)
    s_ref = (rows.float().cpu() ** 2).view(B, H // 32, 32).sum(-1).T
    assert torch.allclose(ssq[:, :B].cpu(), s_ref, rtol=1e-5) and bool((ssq[:, B:] == -1).all())
    q_ref, s_bytes = mx_ref.mx_quant(w_ref)
    assert torch.equal(xqs.cpu(), s_bytes), "E8M0 block scales"
    assert torch.equal(xq.cpu(), q_ref), "e4m3 codes"


FP8_SHAPES = [(4608, 3584, 0, "qkv"), (37888, 3584, 1, "gate/up"), (3584, 18944, 2, "down"), (152064, 3584, 0, "lm_head f32"),
              (512, 256, 0, "tiny qkv"), (256, 768, 2, "tiny down (K padded)"), (1408, 256, 1, "tiny gate/up")]


@pytest.mark.parametrize("form", ["streamk", "colpar"])
@pytest.mark.parametrize("N,K,mode,what", FP8_SHAPES)
@pytest.mark.parametrize("B", [1, 4, 20, 64])
def test_decode_proj_fp8_mx(hip, device, N, K, mode, what, B, form):
    """fp8 form: e4m3 weights with per-row scales, MX activation blocks with NON-trivial block scales (rows and blocks of
    very different magnitude: a wrong scale-to-lane mapping cannot hide) against the fp32 product of the de-quantised operands;
    MX outputs byte-exact against oracle/mx_ref.py applied to the kernel's own bf16 outputs."""
    if N > 100000 and B not in (4, 64):
        pytest.skip("large-N shapes at two batch sizes only")
    seed = 7 * N + K + B
    g = torch.Generator().manual_seed(seed)
    xf = torch.randn((B, K), generator=g) * torch.pow(2.0, torch.randint(-6, 7, (B, K // 32, 1), generator=g).float()).expand(
        B, K // 32, 32).reshape(B, K)
    xq, xs = mx_ref.mx_quant(xf)
    xd = mx_ref.mx_dequant(xq, xs)
    w = _mk((N, K), seed + 1, K ** -0.5)
    wq, sw = hip.quantize_fp8_rows(w.to(device))
    wdq = wq.cpu().view(torch.float8_e4m3fn).float() * sw.cpu()[:, None]
    f32_out = what.endswith("f32")
    bias = _mk((N,), seed + 2, 0.5) if mode == 0 and not f32_out else None
    res = _mk((B, N), seed + 3) if mode == 2 else None
    nw = (1 + 0.1 * _mk((N,), seed + 4).float()).to(torch.bfloat16) if mode == 2 else None
    use_rs = mode != 2
    tiles_in = 3
    ssq_in = (torch.rand((tiles_in, 64), generator=g) * 200 + 20) if use_rs else None
    rs = torch.rsqrt(ssq_in[:, :B].sum(0) / 384 + 1e-6) if use_rs else torch.ones(B)
    main, weighted, _ = _ref_epilogue((xd.to(device) @ wdq.to(device).T).cpu(), mode, rs, bias, res, nw)
    if not _covered(hip, form, N, mode, mode != 0):
        pytest.skip("shape / output outside the column-parallel form")
    n_out = N // 2 if mode == 1 else N
    ws = hip.decode_proj_ws(device, B, N, K, fp8=True)
    out = torch.full((B, n_out), 7.0, dtype=torch.float32 if f32_out else torch.bfloat16, device=device)
    out_w = torch.full((B, n_out), 7.0, dtype=torch.bfloat16, device=device) if mode == 2 else None
    want_q = mode != 0
    oq = torch.zeros((B, n_out), dtype=torch.uint8, device=device) if want_q else None
    oqs = torch.zeros((B, n_out // 32), dtype=torch.uint8, device=device) if want_q else None
    ssq_out = torch.zeros(((N + 31) // 32, 64), dtype=torch.float32, device=device) if mode == 2 else None
    hip.decode_proj_fp8(xq.to(device), xs.to(device), wq, sw, ws, mode, out=out, out_w=out_w, out_q=oq, out_qs=oqs,
                        bias=bias.to(device) if bias is not None else None, residual=res.to(device) if res is not None else None,
                        norm_w=nw.to(device) if nw is not None else None, ssq_in=ssq_in.to(device) if use_rs else None,
                        ssq_out=ssq_out, norm_dim=384 if use_rs else 0, eps=1e-6, form=form)
    torch.cuda.synchronize()
    got = out.float().cpu()
    scale = float(main.abs().max())
    err = float((got - main).abs().max())
    assert err <= (2e-2 if not f32_out else 2e-3) * max(1.0, scale), f"{what} B={B}: max err {err} (range {scale})"
    if want_q:
        consumed = out_w.float().cpu() if mode == 2 else got
        q_ref, s_ref = mx_ref.mx_quant(consumed)
        assert torch.equal(oqs.cpu(), s_ref), "E8M0 scales of the MX output"
        assert torch.equal(oq.cpu(), q_ref), "e4m3 codes of the MX output"


def test_decode_proj_argument_checks(hip, device):
    x, w = _mk((4, 256), 1).to(device), _mk((512, 256), 2).to(device)
    ws = hip.decode_proj_ws(device, 4, 512, 256)
    out = torch.empty((4, 512), dtype=torch.bfloat16, device=device)
    with pytest.raises(hip.HipLibraryError):                       # residual mode without its operands
        hip.decode_proj(x, w, ws, hip.DP_RESID_NORMW, out=out)
    with pytest.raises(hip.HipLibraryError):                       # workspace of another (smaller) shape (stream-K form)
        hip.decode_proj(x, w, ws[:20000], hip.DP_PLAIN, out=out, form="streamk")
    hip.decode_proj(x, w, None, hip.DP_PLAIN, out=out, form="colpar")       # the column-parallel form needs none
    with pytest.raises(hip.HipLibraryError):                       # N outside the column-parallel form
        hip.decode_proj(x, w[:500], None, hip.DP_PLAIN, out=out[:, :500], form="colpar")
    with pytest.raises(hip.HipLibraryError):                       # K not a multiple of the K-step
        hip.decode_proj(x[:, :200], w[:, :200], ws, hip.DP_PLAIN, out=out)
