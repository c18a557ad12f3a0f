"""TEST INFRASTRUCTURE (never imported by the product): numpy restatement of libjpeg-turbo's default decoder AFTER entropy
decoding - dequantisation, the integer "islow" inverse DCT (jidctint.c: jpeg_idct_islow, CONST_BITS 13, PASS1_BITS 2),
"fancy" triangle chroma upsampling (jdsample.c: h2v1_fancy_upsample / h2v2_fancy_upsample, plain replication when the
chroma plane is at most 2 samples wide) and the 16-bit fixed-point YCbCr -> RGB tables (jdcolor.c: build_ycc_rgb_table).

Pinned (tests/test_jpeg.py) to PIL's decode of the same bytes - PIL links libjpeg-turbo and is what the reference's service
side amounts to (the reference itself only ever WRITES JPEGs: src/agents/vlm_inspector.py:46-88) - bit for bit, on
4:2:0 / 4:2:2 / 4:4:4 / grey images of odd and even sizes.  The HIP kernels (csrc/jpeg.hip) are checked against this."""
import numpy as np

FIX_0_298631336, FIX_0_390180644, FIX_0_541196100, FIX_0_765366865 = 2446, 3196, 4433, 6270
FIX_0_899976223, FIX_1_175875602, FIX_1_501321110, FIX_1_847759065 = 7373, 9633, 12299, 15137
FIX_1_961570560, FIX_2_053119869, FIX_2_562915447, FIX_3_072711026 = 16069, 16819, 20995, 25172
CONST_BITS, PASS1_BITS = 13, 2


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(v, shift):
    """v: [..., 8] int64 along the last axis -> the eight outputs of one islow pass, descaled by ``shift``."""
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * FIX_0_541196100
    tmp2 = z1 - z3 * FIX_1_847759065
    tmp3 = z1 + z2 * FIX_0_765366865
    z2, z3 = v[..., 0], v[..., 4]
    tmp0 = (z2 + z3) << CONST_BITS
    tmp1 = (z2 - z3) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * FIX_1_175875602
    tmp0 = tmp0 * FIX_0_298631336
    tmp1 = tmp1 * FIX_2_053119869
    tmp2 = tmp2 * FIX_3_072711026
    tmp3 = tmp3 * FIX_1_501321110
    z1 = -z1 * FIX_0_899976223
    z2 = -z2 * FIX_2_562915447
    z3 = -z3 * FIX_1_961570560 + z5
    z4 = -z4 * FIX_0_390180644 + z5
    tmp0 = tmp0 + z1 + z3
    tmp1 = tmp1 + z2 + z4
    tmp2 = tmp2 + z2 + z3
    tmp3 = tmp3 + z1 + z4
    out = [tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3]
    return np.stack([_descale(o, shift) for o in out], axis=-1)


def idct_blocks(coeffs, qt):
    """coeffs [n, 64] int16 (natural order, quantised), qt [64] -> [n, 8, 8] uint8 samples."""
    c = coeffs.astype(np.int64).reshape(-1, 8, 8) * np.asarray(qt, dtype=np.int64).reshape(8, 8)
    ws = _idct_1d(c.transpose(0, 2, 1), CONST_BITS - PASS1_BITS).transpose(0, 2, 1)      # pass 1: columns
    out = _idct_1d(ws, CONST_BITS + PASS1_BITS + 3)                                      # pass 2: rows
    out = ((out + 512) & 1023) - 512                                                     # range_limit[x & RANGE_MASK]
    return np.clip(out + 128, 0, 255).astype(np.uint8)


def plane(coeffs, qt, bh, bw):
    s = idct_blocks(coeffs, qt).reshape(bh, bw, 8, 8)
    return s.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def upsample_h2(p, dw):
    """h2v1 fancy: [rows, >= dw] -> [rows, 2 dw] (int64 in, int64 out; replication when dw <= 2)."""
    p = p[:, :dw].astype(np.int64)
    if dw <= 2:
        return np.repeat(p, 2, axis=1)
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
    out = np.empty((p.shape[0], 2 * dw), dtype=np.int64)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out


def upsample_h2v2(p, dw, dh):
    """h2v2 fancy: the real dh x dw samples of a chroma plane -> [2 dh, 2 dw] (replication when dw <= 2)."""
    p = p[:dh, :dw].astype(np.int64)
    if dw <= 2:
        return np.repeat(np.repeat(p, 2, axis=0), 2, axis=1)
    above = np.concatenate([p[:1], p[:-1]], axis=0)
    below = np.concatenate([p[1:], p[-1:]], axis=0)
    out = np.empty((2 * dh, 2 * dw), dtype=np.int64)
    for v, far in ((0, above), (1, below)):
        cs = 3 * p + far                                      # "colsum" of the nearer and the further row
        left = np.concatenate([cs[:, :1], cs[:, :-1]], axis=1)
        right = np.concatenate([cs[:, 1:], cs[:, -1:]], axis=1)
        out[v::2, 0::2] = (3 * cs + left + 8) >> 4
        out[v::2, 1::2] = (3 * cs + right + 7) >> 4
    return out


def ycc_to_rgb(y, cb, cr):
    ONE_HALF, S = 1 << 15, 16
    fix = lambda x: int(x * (1 << S) + 0.5)
    cbx, crx = cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((fix(1.40200) * crx + ONE_HALF) >> S)
    b = y + ((fix(1.77200) * cbx + ONE_HALF) >> S)
    g = y + ((-fix(0.34414) * cbx + ONE_HALF - fix(0.71414) * crx) >> S)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def decode(info, coeffs):
    """info: dict with width, height, ncomp, hs, vs, bw, bh, dw, dh, qt; coeffs [total_blocks, 64] int16 -> RGB uint8."""
    W, H = info["width"], info["height"]
    planes, o = [], 0
    for c in range(info["ncomp"]):
        n = info["bw"][c] * info["bh"][c]
        planes.append(plane(coeffs[o:o + n], info["qt"][c], info["bh"][c], info["bw"][c]))
        o += n
    y = planes[0][:H, :W].astype(np.int64)
    if info["ncomp"] == 1:
        return np.repeat(planes[0][:H, :W, None], 3, axis=2)
    h, v = info["hs"][0], info["vs"][0]
    ch = []
    for c in (1, 2):
        if (h, v) == (1, 1):
            up = planes[c].astype(np.int64)
        elif (h, v) == (2, 1):
            up = upsample_h2(planes[c][:info["dh"][c]], info["dw"][c])
        else:
            up = upsample_h2v2(planes[c], info["dw"][c], info["dh"][c])
        ch.append(up[:H, :W])
    return ycc_to_rgb(y, ch[0], ch[1])
