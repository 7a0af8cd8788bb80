"""LDS-tiled depthwise kernels micro-benchmark (GPU box): forward and fused backward on the distinct stride-1 MBConv shapes of
EfficientNet-B4 at B = 256; time, effective GB/s over the algorithmic bytes (fwd: in + out; bwd: dy + z2 + z1 + out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
from multimodalsimilar_amd._lib import lib
s = ops._stream()
B = 256
SH = [(56, 192, 3, 3), (28, 336, 5, 3), (14, 672, 3, 5), (14, 960, 5, 5), (7, 1632, 5, 7), (7, 2688, 3, 1), (112, 48, 3, 1)]
scr = torch.empty(32 << 20, device="cuda")
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot_f = tot_b = 0.0
for H, C, K, cnt in SH:
    P = B * H * H
    z1 = torch.randn(P, C, device="cuda").half(); z2 = torch.randn(P, C, device="cuda").half(); dy = torch.randn(P, C, device="cuda").bfloat16()      # forward tensors fp16, gradients bf16
    out = torch.empty(P, C, dtype=torch.bfloat16, device="cuda")
    mk = lambda: (torch.randn(C, device="cuda") * 0.1, 1 + 0.1 * torch.rand(C, device="cuda"), 1 + 0.1 * torch.randn(C, device="cuda"), 0.1 * torch.randn(C, device="cuda"))
    mu1, rs1, sc1, sh1 = mk(); mu2, rs2, sc2, sh2 = mk()
    wT = torch.randn(K * K, C, device="cuda") * 0.2
    sums = torch.zeros(2 * C, device="cuda"); sums2 = torch.randn(2 * C, device="cuda"); sums1 = torch.zeros(2 * C, device="cuda")
    gate = torch.rand(B, C, device="cuda"); dsq = torch.randn(B, C, device="cuda")
    gT = torch.zeros(K * K, C, device="cuda"); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    plain = C == 48
    fw = lambda: lib.dwtile_fwd(z1.data_ptr(), None if plain else sc1.data_ptr(), None if plain else sh1.data_ptr(), wT.data_ptr(), out.data_ptr(), sums.data_ptr(), B, H, H, C, K, 1, scr.data_ptr(), scr.numel(), s)
    bw = lambda: lib.dwtile_bwd(dy.data_ptr(), z2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), mu2.data_ptr(), rs2.data_ptr(), sums2.data_ptr(), gate.data_ptr(), dsq.data_ptr(), z1.data_ptr(),
                                *( [None] * 4 if plain else [sc1.data_ptr(), sh1.data_ptr(), mu1.data_ptr(), rs1.data_ptr()]), None, wT.data_ptr(), out.data_ptr(), None if plain else sums1.data_ptr(), gT.data_ptr(), dg.data_ptr(), db.data_ptr(), B, H, H, C, K, scr.data_ptr(), scr.numel(), s)
    tf, tb = t(fw), t(bw)
    by = P * C * 2
    extra = ""
    if K == 5 and lib.dw5m_eligible(B, H, H, C, 5, 1):          # the matrix-core kernels (csrc/dwmfma.hip) on the same shape
        fm = lambda: lib.dw5m_fwd(z1.data_ptr(), sc1.data_ptr(), sh1.data_ptr(), wT.data_ptr(), out.data_ptr(), sums.data_ptr(), B, H, H, C, scr.data_ptr(), scr.numel(), s)
        tfm = t(fm)
        extra = f" || dw5m fwd {tfm:7.1f} us {2 * by / tfm / 1e3:5.0f} GB/s"
        if hasattr(lib, "_decls") and "mmsim_dw5m_bwd" in lib._decls:
            bm = lambda: lib.dw5m_bwd(dy.data_ptr(), z2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), mu2.data_ptr(), rs2.data_ptr(), sums2.data_ptr(), gate.data_ptr(), dsq.data_ptr(), z1.data_ptr(),
                                      sc1.data_ptr(), sh1.data_ptr(), mu1.data_ptr(), rs1.data_ptr(), wT.data_ptr(), out.data_ptr(), sums1.data_ptr(), gT.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                      B, H, H, C, scr.data_ptr(), scr.numel(), s)
            tbm = t(bm)
            extra += f" | bwd {tbm:7.1f} us {4 * by / tbm / 1e3:5.0f} GB/s"
    print(f"x{cnt} {H:3d}^2 C={C:4d} k{K}: fwd {tf:7.1f} us {2 * by / tf / 1e3:5.0f} GB/s | bwd {tb:7.1f} us {4 * by / tb / 1e3:5.0f} GB/s{extra}", flush=True)
    tot_f += tf * cnt; tot_b += tb * cnt
print(f"total fwd {tot_f / 1e3:.2f} ms, bwd {tot_b / 1e3:.2f} ms per step (stride-1 blocks)")
