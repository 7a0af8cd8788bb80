"""The matrix-core depthwise kernels alone (csrc/dwmfma.hip) on the k5 stride-1 shapes of EfficientNet-B4 at B = 256: time per launch
and GB/s over the algorithmic bytes; used under rocprofv3 (tools/pmc_dwm.sh) and for A/B of variant builds (MMSIM_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
from multimodalsimilar_amd._lib import lib
s = ops._stream()
B = 256
SH = [(28, 336), (14, 960), (7, 1632)]
scr = torch.empty(32 << 20, device="cuda")
NREP = int(os.environ.get("NREP", 5))
def t(f, n=NREP):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out_l = []
for H, C in SH:
    P = B * H * H
    z1 = torch.randn(P, C, device="cuda").half(); out = torch.empty(P, C, dtype=torch.float16, device="cuda")
    sc1 = 1 + 0.1 * torch.randn(C, device="cuda"); sh1 = 0.1 * torch.randn(C, device="cuda")
    wT = torch.randn(25, C, device="cuda") * 0.2
    sums = torch.zeros(2 * C, device="cuda")
    fm = lambda: lib.dw5m_fwd(z1.data_ptr(), sc1.data_ptr(), sh1.data_ptr(), wT.data_ptr(), out.data_ptr(), sums.data_ptr(), B, H, H, C, scr.data_ptr(), scr.numel(), s)
    tf = t(fm)
    z2 = torch.randn(P, C, device="cuda").half(); dy = (torch.randn(P, C, device="cuda") * 0.1).bfloat16(); dout = torch.empty(P, C, dtype=torch.bfloat16, device="cuda")
    mu1 = torch.randn(C, device="cuda") * 0.1; rs1 = 1 + 0.1 * torch.rand(C, device="cuda"); mu2 = torch.randn(C, device="cuda") * 0.1; rs2 = 1 + 0.1 * torch.rand(C, device="cuda")
    sc2 = 1 + 0.1 * torch.randn(C, device="cuda"); sh2 = 0.1 * torch.randn(C, device="cuda"); sums2 = torch.randn(2 * C, device="cuda") * 0.1; sums1 = torch.zeros(2 * C, device="cuda")
    gate = torch.rand(B, C, device="cuda"); dsq = torch.randn(B, C, device="cuda") * 0.05; gT = torch.zeros(25, C, device="cuda"); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    bm = lambda: lib.dw5m_bwd(dy.data_ptr(), z2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), mu2.data_ptr(), rs2.data_ptr(), sums2.data_ptr(), gate.data_ptr(), dsq.data_ptr(), z1.data_ptr(),
                              sc1.data_ptr(), sh1.data_ptr(), mu1.data_ptr(), rs1.data_ptr(), wT.data_ptr(), dout.data_ptr(), sums1.data_ptr(), gT.data_ptr(), dg.data_ptr(), db.data_ptr(),
                              B, H, H, C, scr.data_ptr(), scr.numel(), s)
    tb = t(bm) if os.environ.get("DWM_BWD", "1") != "0" else float("nan")
    out_l.append(f"{H}^2x{C}: fwd {tf:6.1f} us {2 * P * C * 2 / tf / 1e3:5.0f} GB/s bwd {tb:6.1f} us {4 * P * C * 2 / tb / 1e3:5.0f} GB/s")
print(" | ".join(out_l), flush=True)
