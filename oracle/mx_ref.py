"""TEST INFRASTRUCTURE (oracle): CPU statement of the MX-style activation blocks the round-5 batched decode uses in the fp8
configuration (BASELINE configs[4]) - OCP e4m3 elements with one E8M0 (power of two) scale per row and 32 consecutive
columns, as consumed by v_mfma_scale_f32_16x16x128_f8f6f4 (one scale per lane = per 32-element block).

The reference (Aditya-Somasi/Vision-Inspection-System) holds no arithmetic (remote API); the element format
follows the OCP 8-bit floating point / Microscaling specifications (e4m3 "fn": max 448, no infinities; E8M0: 2^(byte - 127)).
The scale rule is this repository's (csrc/decode_stream.hip ds_mx_scale_byte): the SMALLEST power of two X with
block_max / X <= 448, so no element saturates.  Only tests import this module; the product never does."""
import torch


def mx_scale_bytes(amax: torch.Tensor) -> torch.Tensor:
    """E8M0 bytes for block maxima (f32 tensor): E - 8 when the mantissa of amax is <= 1.75 (448 = 1.75 * 2^8), else E - 7."""
    bits = amax.to(torch.float32).contiguous().view(torch.int32)
    e = (bits >> 23) & 0xFF
    man = bits & 0x7FFFFF
    b = e - torch.where(man <= 0x600000, torch.full_like(e, 8), torch.full_like(e, 7))
    return b.clamp(0, 254).to(torch.uint8)


def mx_quant(u: torch.Tensor):
    """u [B, n] f32 (n % 32 == 0) -> (e4m3 bytes [B, n] uint8, E8M0 scales [B, n / 32] uint8)."""
    B, n = u.shape
    blocks = u.to(torch.float32).reshape(B, n // 32, 32)
    sb = mx_scale_bytes(blocks.abs().amax(-1))
    inv = torch.pow(2.0, (127 - sb.to(torch.int32)).to(torch.float32))
    q = (blocks * inv[..., None]).to(torch.float8_e4m3fn).view(torch.uint8).reshape(B, n)
    return q, sb


def mx_dequant(q: torch.Tensor, sb: torch.Tensor) -> torch.Tensor:
    """Inverse of mx_quant up to the e4m3 rounding: f32 [B, n]."""
    B, n = q.shape
    v = q.view(torch.float8_e4m3fn).to(torch.float32).reshape(B, n // 32, 32)
    sc = torch.pow(2.0, (sb[:, :n // 32].to(torch.int32) - 127).to(torch.float32))
    return (v * sc[..., None]).reshape(B, n)


def fake_quant_mx_e4m3(x: torch.Tensor) -> torch.Tensor:
    """quantise -> de-quantise of the last dimension in MX blocks (zero-padded to a multiple of 32)."""
    shape = x.shape
    x2 = x.reshape(-1, shape[-1]).to(torch.float32)
    n = x2.shape[1]
    pad = (-n) % 32
    if pad:
        x2 = torch.cat([x2, torch.zeros((x2.shape[0], pad), dtype=x2.dtype)], 1)
    q, sb = mx_quant(x2)
    return mx_dequant(q, sb)[:, :n].reshape(shape)
