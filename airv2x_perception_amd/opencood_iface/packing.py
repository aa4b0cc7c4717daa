"""Host-side weight preparation for the C-ABI kernels (done once per state_dict).

* eval-mode BatchNorm (eps 1e-3 everywhere on the path: airv2x_pillar_vfe.py:21,
  base_bev_backbone.py:52,65,83) is folded into a per-channel (scale, shift) pair computed in
  float64 and rounded once to fp32;
* convolution weights are re-laid out for the implicit-GEMM kernel:
  ``[tap][cin/4][coutp][4]`` (include/airv2x_hip.h, av2x_conv2d).
"""
from __future__ import annotations

import torch

BN_EPS = 1e-3


def fold_bn(sd, prefix, eps=BN_EPS):
    w = sd[prefix + ".weight"].detach().double().cpu()
    b = sd[prefix + ".bias"].detach().double().cpu()
    m = sd[prefix + ".running_mean"].detach().double().cpu()
    v = sd[prefix + ".running_var"].detach().double().cpu()
    scale = w / torch.sqrt(v + eps)
    shift = b - m * scale
    return scale.float().contiguous(), shift.float().contiguous()


def round_up(x, m):
    return (x + m - 1) // m * m


def pack_conv_weight(w):
    """(Cout, Cin, k, k) -> (k*k, Cin/4, CoutP, 4) fp32 contiguous, CoutP = Cout rounded up to 32."""
    w = w.detach().float().cpu()
    cout, cin, kh, kw = w.shape
    assert cin % 4 == 0
    coutp = round_up(cout, 32)
    t = w.permute(2, 3, 1, 0).reshape(kh * kw, cin // 4, 4, cout).permute(0, 1, 3, 2)
    out = torch.zeros(kh * kw, cin // 4, coutp, 4, dtype=torch.float32)
    out[:, :, :cout, :] = t
    return out.contiguous(), coutp


def pack_deconv_weight(w):
    """ConvTranspose2d weight (Cin, Cout, s, s), kernel == stride ->
    (1, Cin/4, s*s*Cout, 4); GEMM column n = (i*s + j)*Cout + co."""
    w = w.detach().float().cpu()
    cin, cout, s, s2 = w.shape
    assert s == s2 and cin % 4 == 0 and cout % 32 == 0
    ncol = s * s * cout
    t = w.permute(0, 2, 3, 1).reshape(cin // 4, 4, ncol).permute(0, 2, 1)
    return t.reshape(1, cin // 4, ncol, 4).contiguous(), ncol


def to_bf16_koct(packed):
    """fp32 k-quad packing (T, Cin/4, CoutP, 4) of pack_conv_weight / pack_deconv_weight -> the bf16 k-oct packing
    (T, Cin/8, CoutP, 8) read by conv_igemm_bf16 (tile flag 0x0800): element [t][o][n][e] = W[k = 8 o + e][n],
    rounded to bf16 (round-to-nearest-even, as torch.autocast does)."""
    T, q, n, four = packed.shape
    assert four == 4 and q % 2 == 0
    t = packed.reshape(T, q // 2, 2, n, 4).permute(0, 1, 3, 2, 4).reshape(T, q // 2, n, 8)
    return t.to(torch.bfloat16).contiguous()


def interleave2_columns(w16):
    """bf16 k-oct packing (1, K/8, CoutP, 8) -> the column order av2x_linear_bf16 reads (csrc/linear_bf16.hip): CoutP padded
    with zero columns to a multiple of 256, and inside every group of 64 columns packed column 32 c + i = logical column
    2 i + c, so that an MFMA lane (i) owns the two consecutive logical columns 2 i, 2 i + 1 over its two tiles (c)."""
    T, q, n, eight = w16.shape
    assert T == 1 and eight == 8
    npad = (n + 255) // 256 * 256
    if npad != n:
        w16 = torch.cat([w16, torch.zeros(T, q, npad - n, 8, dtype=w16.dtype, device=w16.device)], 2)
    t = w16.reshape(T, q, npad // 64, 32, 2, 8).permute(0, 1, 2, 4, 3, 5)
    return t.reshape(T, q, npad, 8).contiguous(), npad


def to_bf16x3_koct(packed):
    """Split-3 packing for conv_igemm_bf16x3 (tile flag 0x0400): (3, T, Cin/8, CoutP, 8) bf16 planes hi / mid / lo with
    hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid) (each subtraction is exact in fp32), so hi + mid + lo = w to 2^-24."""
    T, q, n, four = packed.shape
    assert four == 4 and q % 2 == 0
    w = packed.reshape(T, q // 2, 2, n, 4).permute(0, 1, 3, 2, 4).reshape(T, q // 2, n, 8).float()
    hi = w.to(torch.bfloat16)
    r1 = w - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return torch.stack([hi, mid, lo]).contiguous()
