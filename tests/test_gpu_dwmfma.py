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
    check(tag, "output relative L2 vs the VALU tile kernel", (z1n - z0).norm() / z0.norm(), 3e-4)
    check(tag, "max |difference| / max |output|", (z1n - z0).abs().max() / z0.abs().max(), 2e-3)      # one fp16 ulp of a large output
    check(tag, "statistics relative L2", (s1n - s0).norm() / s0.norm(), 1e-3)
