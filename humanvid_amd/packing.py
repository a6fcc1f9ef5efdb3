"""One-time weight re-layout for the gfx950 kernels (load time, not on the step path).

The nn.Module tree keeps the reference's parameter names and shapes (state-dict compatible);
these helpers derive the device-side operands from it:
  * 3x3 conv weights [Cout, Cin, 3, 3] -> [Cout, 9, Cin_pad] bf16 (tap-major, channel-minor: the
    reduction order of hv_conv3x3),
  * LayerNorm folded into the following Linear: W' = bf16(W * gamma), colsum[n] = sum_k W'[n,k],
    bias'[n] = bias[n] + sum_k beta[k] W[n,k]   (hv_gemm epilogue: rstd*(acc - mean*colsum) + bias'),
  * sinusoidal positional encoding folded through the projection: pe_table[f, n] = sum_k pe[f,k] W[n,k],
  * GEGLU projection rows interleaved in blocks of 16 [h | g] so the gate is applied in-register.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

BF16 = torch.bfloat16


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pack_conv3x3(w: torch.Tensor, cin_pad: Optional[int] = None) -> torch.Tensor:
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    cin_pad = cin_pad or round_up(cin, 32)
    out = torch.zeros(cout, 9, cin_pad, dtype=BF16, device=w.device)
    out[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, 9, cin).to(BF16)
    return out.contiguous()


def pack_conv3x3_two_source(w: torch.Tensor, c1: int) -> torch.Tensor:
    """Weights of a conv reading cat([x1 (c1 ch), x2], dim=channel): same order, no padding needed."""
    return pack_conv3x3(w, w.shape[1])


def pack_linear(w: torch.Tensor, k_pad: Optional[int] = None) -> torch.Tensor:
    """[N, K] (nn.Linear) or [N, K, 1, 1] (1x1 conv) -> [N, K_pad] bf16."""
    w = w.reshape(w.shape[0], -1)
    n, k = w.shape
    k_pad = k_pad or round_up(k, 64)
    out = torch.zeros(n, k_pad, dtype=BF16, device=w.device)
    out[:, :k] = w.to(BF16)
    return out.contiguous()


def fold_layernorm(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor
                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    w32 = w.float()
    wf = (w32 * gamma.float()[None, :]).to(BF16)
    colsum = wf.float().sum(dim=1)
    b = w32 @ beta.float()
    if bias is not None:
        b = b + bias.float()
    return wf.contiguous(), colsum.contiguous(), b.contiguous()


def pe_table(pe: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """pe [F, K], w [N, K] (un-folded weights) -> [F, N] fp32."""
    return (pe.float() @ w.float().t()).contiguous()


def geglu_row_order(n2: int, device=None) -> torch.Tensor:
    """Row permutation for a GEGLU projection with 2*inner rows ([h ; g] halves): blocks of
    16 h-rows followed by the matching 16 g-rows."""
    inner = n2 // 2
    assert inner % 16 == 0
    idx = torch.arange(inner, device=device).view(-1, 16)
    return torch.cat([idx, idx + inner], dim=1).reshape(-1)


def pack_geglu(w: torch.Tensor, bias: torch.Tensor):
    order = geglu_row_order(w.shape[0], w.device)
    return w[order].contiguous(), bias[order].contiguous(), order
