"""Weight containers and layout transforms for the gfx950 kernels.

Layout decisions (DESIGN.md "Data layout in HBM"):
  * every nn.Linear weight stays [out, in] (K-contiguous) in bf16 - the "NT" GEMM operand;
  * q/k/v projections are concatenated into one [Hq*D + 2*Hkv*D, hidden] matrix (one GEMM);
  * gate/up projections are interleaved in 16-row groups ([gate 16 | up 16] ...) so the
    SwiGLU product happens in the GEMM/GEMV epilogue (same lane holds gate_i and up_i);
  * the patch-embed Conv3d weight [E, 3, 2, 14, 14] is flattened to [E, 1176] and zero-padded
    to 1216 columns (a multiple of the GEMM K-step of 64).
"""
from __future__ import annotations

import torch


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[I,K],[I,K] -> [2I,K] with rows [g0..g15, u0..u15, g16..g31, u16..u31, ...]. I % 16 == 0."""
    I, K = gate.shape
    if up.shape != gate.shape or I % 16 != 0:
        raise ValueError("gate/up must have equal shapes with rows % 16 == 0")
    g = gate.reshape(I // 16, 16, K)
    u = up.reshape(I // 16, 16, K)
    return torch.stack((g, u), dim=1).reshape(2 * I, K).contiguous()


def pad_cols(w: torch.Tensor, multiple: int) -> torch.Tensor:
    """Zero-pad the last dimension of a 2-D tensor up to a multiple."""
    n, k = w.shape
    kp = (k + multiple - 1) // multiple * multiple
    if kp == k:
        return w.contiguous()
    out = torch.zeros((n, kp), dtype=w.dtype, device=w.device)
    out[:, :k] = w
    return out
