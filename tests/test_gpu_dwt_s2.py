"""The stride-2 form of the fused depthwise backward (mmsim_dwtile_bwd_s2, csrc/mbconv.hip dwt_bwd_kernel<K, false, 2>) against fp32 torch
autograd of the same arithmetic: timm's conv_dw with stride 2 and symmetric padding K/2 (the first block of EfficientNet stages 2, 3, 4, 6
under cv_classifier.py:49) between bn1 + SiLU and bn2 + SiLU + SE gate.  Even and odd planes, planes smaller than a tile, channel counts
that are not a multiple of the block's channel group."""
import pytest
import torch
import torch.nn.functional as F

from parity_log import check

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _silu_grad(u):
    sg = torch.sigmoid(u)
    return sg * (1 + u * (1 - sg))


@pytest.mark.parametrize("B,H,W,C,K", [(2, 16, 16, 16, 3), (3, 14, 14, 24, 5), (2, 15, 13, 40, 3), (1, 7, 9, 8, 5), (2, 56, 56, 144, 3),
                                       (2, 28, 28, 336, 3), (3, 14, 14, 960, 5), (1, 112, 112, 24, 3)])
def test_stride2_fused_backward_vs_torch(B, H, W, C, K):
    from multimodalsimilar_amd import ops
    from multimodalsimilar_amd._lib import lib
    torch.manual_seed(B * 100 + H + K)
    s = ops._stream()
    pad = K // 2
    Ho, Wo = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
    Pi, Po = B * H * W, B * Ho * Wo
    z1 = torch.randn(Pi, C, device=DEV).half(); z2 = torch.randn(Po, C, device=DEV).half(); dy = (torch.randn(Po, C, device=DEV) * 0.1).bfloat16()
    mk = lambda: (torch.randn(C, device=DEV) * 0.1, 1 + 0.1 * torch.rand(C, device=DEV), 1 + 0.1 * torch.randn(C, device=DEV), 0.1 * torch.randn(C, device=DEV))
    mu1, rs1, sc1, sh1 = mk(); mu2, rs2, sc2, sh2 = mk()
    wT = torch.randn(K * K, C, device=DEV) * 0.2
    sums2 = torch.randn(2 * C, device=DEV) * 0.1
    gate = torch.rand(B, C, device=DEV); dsq = torch.randn(B, C, device=DEV) * 0.05
    scr = torch.empty(16 << 20, device=DEV)
    out = torch.full((Pi, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    sums1 = torch.zeros(2 * C, device=DEV); gT = torch.zeros(K * K, C, device=DEV); dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
    lib.dwtile_bwd_s2(dy.data_ptr(), z2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), mu2.data_ptr(), rs2.data_ptr(), sums2.data_ptr(),
                      gate.data_ptr(), dsq.data_ptr(), z1.data_ptr(), sc1.data_ptr(), sh1.data_ptr(), mu1.data_ptr(), rs1.data_ptr(),
                      wT.data_ptr(), out.data_ptr(), sums1.data_ptr(), gT.data_ptr(), dg.data_ptr(), db.data_ptr(), B, H, W, C, K,
                      scr.data_ptr(), scr.numel(), s)
    torch.cuda.synchronize()
    # ---- the same arithmetic in fp32 torch
    invP, inv_hw = 1.0 / Po, 1.0 / (Ho * Wo)
    z2f = z2.float().view(B, Ho * Wo, C); dyf = dy.float().view(B, Ho * Wo, C)
    da = (dyf * gate[:, None, :] + dsq[:, None, :] * inv_hw) * _silu_grad(z2f * sc2 + sh2)
    dz2 = sc2 * (da - sums2[:C] * invP - (z2f - mu2) * rs2 * sums2[C:] * invP)
    dz2 = dz2.bfloat16().float()                                    # staged in LDS as bf16
    u1 = z1.float() * sc1 + sh1
    a1 = (u1 * torch.sigmoid(u1)).half().float()                    # the rounded a1 the forward convolved
    a1v = a1.view(B, H, W, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    w = wT.t().reshape(C, 1, K, K).contiguous().requires_grad_(True)
    y = F.conv2d(a1v, w, stride=2, padding=pad, groups=C)
    y.backward(dz2.view(B, Ho, Wo, C).permute(0, 3, 1, 2).contiguous())
    da1 = a1v.grad.permute(0, 2, 3, 1).reshape(Pi, C)
    dpre = (da1 * _silu_grad(u1)).bfloat16().float()
    ref_s = torch.cat([dpre.sum(0), (dpre * (z1.float() - mu1) * rs1).sum(0)])
    ref_g = w.grad.reshape(C, K * K).t()
    o = out.float()
    assert torch.isfinite(o).all() and torch.isfinite(gT).all()
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-20)).item()
    tag = f"dwtile_bwd_s2[{B}x{H}x{W}x{C} k{K}]"
    check(tag, "dpre relative L2 vs torch fp32", rel(o, dpre), 4e-3)
    check(tag, "expand-BatchNorm backward sums relative L2", rel(sums1, ref_s), 8e-3)
    check(tag, "weight gradient relative L2", rel(gT, ref_g), 4e-3)
    assert torch.allclose(dg, sums2[C:]) and torch.allclose(db, sums2[:C])
