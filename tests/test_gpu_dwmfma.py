"""The matrix-core 5 x 5 depthwise kernels (csrc/dwmfma.hip) against the LDS-tiled VALU kernels they replace (csrc/mbconv.hip, themselves
held to the oracle by tests/test_gpu_image_tower.py) on the same random data: shapes with ragged planes (13 x 13), batch sizes that are
not a multiple of the images per tile, the production planes (28, 14, 7).  timm conv_dw k = 5, s = 1 under cv_classifier.py:49."""
import pytest
import torch

from parity_log import check

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scr():
    t = torch.empty(16 << 20, device=DEV)
    return t


@pytest.mark.parametrize("B,H,C,xf", [(3, 28, 48, True), (5, 14, 64, True), (9, 7, 32, True), (2, 13, 16, True), (4, 14, 32, False), (33, 7, 1632, True)])
def test_dw5m_forward_equals_the_tile_kernel(B, H, C, xf):
    from multimodalsimilar_amd import ops
    from multimodalsimilar_amd._lib import lib
    torch.manual_seed(B * 100 + H)
    s = ops._stream()
    P = B * H * H
    z1 = torch.randn(P, C, device=DEV).half()
    sc = (1 + 0.2 * torch.randn(C, device=DEV)) if xf else None
    sh = (0.3 * torch.randn(C, device=DEV)) if xf else None
    wT = torch.randn(25, C, device=DEV) * 0.2
    scr = _scr()
    outs = []
    for fn in ("old", "new"):
        z = torch.full((P, C), float("nan"), dtype=torch.float16, device=DEV)
        sums = torch.zeros(2 * C, device=DEV)
        a = (z1.data_ptr(), sc.data_ptr() if xf else None, sh.data_ptr() if xf else None, wT.data_ptr(), z.data_ptr(), sums.data_ptr(), B, H, H, C)
        if fn == "old":
            lib.dwtile_fwd(*a, 5, 1, scr.data_ptr(), scr.numel(), s)
        else:
            assert lib.dw5m_eligible(B, H, H, C, 5, 1)
            lib.dw5m_fwd(*a, scr.data_ptr(), scr.numel(), s)
        torch.cuda.synchronize()
        outs.append((z.float(), sums.clone()))
    (z0, s0), (z1n, s1n) = outs
    assert torch.isfinite(z1n).all()
    tag = f"dw5m_fwd[{B}x{H}x{C}{'' if xf else '-plain'}]"
    # the matrix-core kernel reads fp16 taps (as the forward 1x1 convolutions read fp16 weight shadows), the VALU kernel fp32 taps: 2^-12
    # RMS per tap on top of the fp16 rounding of the output both share
    check(tag, "output relative L2 vs the VALU tile kernel", (z1n - z0).norm() / z0.norm(), 6e-4)
    check(tag, "max |difference| / max |output|", (z1n - z0).abs().max() / z0.abs().max(), 3e-3)      # a few fp16 ulps of a large output
    check(tag, "statistics relative L2", (s1n - s0).norm() / s0.norm(), 1e-3)


@pytest.mark.parametrize("B,H,C", [(3, 28, 48), (5, 14, 64), (9, 7, 32), (2, 13, 16), (33, 7, 1632), (4, 28, 336)])
def test_dw5m_backward_equals_the_tile_kernel(B, H, C):
    """dz2 staging, data gradient, expand-BatchNorm backward sums, weight gradient and the depthwise BatchNorm's parameter gradients of
    the matrix-core kernel against dwt_bwd_kernel<5, false> on the same data."""
    from multimodalsimilar_amd import ops
    from multimodalsimilar_amd._lib import lib
    torch.manual_seed(B * 10 + H)
    s = ops._stream()
    P = B * H * H
    z1 = torch.randn(P, C, device=DEV).half(); z2 = torch.randn(P, C, device=DEV).half(); dy = (torch.randn(P, C, device=DEV) * 0.1).bfloat16()
    mk = lambda: (torch.randn(C, device=DEV) * 0.1, 1 + 0.1 * torch.rand(C, device=DEV), 1 + 0.1 * torch.randn(C, device=DEV), 0.1 * torch.randn(C, device=DEV))
    mu1, rs1, sc1, sh1 = mk(); mu2, rs2, sc2, sh2 = mk()
    wT = torch.randn(25, C, device=DEV) * 0.2
    sums2 = torch.randn(2 * C, device=DEV) * 0.1
    gate = torch.rand(B, C, device=DEV); dsq = torch.randn(B, C, device=DEV) * 0.05
    scr = _scr()
    res = []
    for fn in ("old", "new"):
        out = torch.full((P, C), float("nan"), dtype=torch.bfloat16, device=DEV)
        sums1 = torch.zeros(2 * C, device=DEV); gT = torch.zeros(25, C, device=DEV); dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
        a = (dy.data_ptr(), z2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), mu2.data_ptr(), rs2.data_ptr(), sums2.data_ptr(), gate.data_ptr(), dsq.data_ptr(),
             z1.data_ptr(), sc1.data_ptr(), sh1.data_ptr(), mu1.data_ptr(), rs1.data_ptr())
        if fn == "old":
            lib.dwtile_bwd(*a, None, wT.data_ptr(), out.data_ptr(), sums1.data_ptr(), gT.data_ptr(), dg.data_ptr(), db.data_ptr(), B, H, H, C, 5,
                           scr.data_ptr(), scr.numel(), s)
        else:
            lib.dw5m_bwd(*a, wT.data_ptr(), out.data_ptr(), sums1.data_ptr(), gT.data_ptr(), dg.data_ptr(), db.data_ptr(), B, H, H, C,
                         scr.data_ptr(), scr.numel(), s)
        torch.cuda.synchronize()
        res.append((out.float(), sums1.clone(), gT.clone(), dg.clone(), db.clone()))
    (o0, s0, g0, dg0, db0), (o1, s1n, g1, dg1, db1) = res
    assert torch.isfinite(o1).all() and torch.isfinite(g1).all()
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-20)).item()
    tag = f"dw5m_bwd[{B}x{H}x{C}]"
    # bf16 taps in the data gradient (the dgrad 1x1 convolutions read bf16 weight shadows as well) and a1 rounded to bf16 in the weight gradient
    check(tag, "dpre relative L2 vs the VALU tile kernel", rel(o1, o0), 4e-3)
    check(tag, "expand-BatchNorm backward sums relative L2", rel(s1n, s0), 8e-3)       # small planes: few terms, bf16 roundings do not average out
    check(tag, "weight gradient relative L2", rel(g1, g0), 4e-3)
    assert torch.equal(dg1, dg0) and torch.equal(db1, db0)
