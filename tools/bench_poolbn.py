"""mmsim_pool_bn_bwd at the depthwise-output shapes of EfficientNet-B4 @ 224, B = 256: time and TB/s over (z2 fp16 + dy bf16), rotating
operand sets (> 256 MB) so that the reads come from HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
from multimodalsimilar_amd._lib import lib
s = ops._stream()
B = 256
def t(fs, n=5):
    for f in fs: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        for f in fs: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * len(fs)) * 1e3
tot = 0.0
for cnt, hw, C in ((2, 112 * 112, 48), (1, 56 * 56, 144), (3, 56 * 56, 192), (1, 28 * 28, 192), (3, 28 * 28, 336), (1, 14 * 14, 336), (5, 14 * 14, 672), (6, 14 * 14, 960), (1, 7 * 7, 960), (7, 7 * 7, 1632), (2, 7 * 7, 2688)):
    P = B * hw
    nset = max(1, min(4, int(300e6 // (P * C * 4)) + 1))
    z = [torch.randn(P, C, device="cuda").half() for _ in range(nset)]
    dy = [torch.randn(P, C, device="cuda").bfloat16() for _ in range(nset)]
    v = [torch.rand(C, device="cuda") + 0.5 for _ in range(4)]
    out5 = torch.empty(5 * B * C, device="cuda")
    fs = [(lambda i=i: lib.pool_bn_bwd(z[i].data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(), dy[i].data_ptr(), out5.data_ptr(), B, hw, C, s)) for i in range(nset)]
    us = t(fs)
    tot += cnt * us
    print(f"x{cnt} HW={hw:6d} C={C:5d}: {us:7.1f} us  {P * C * 4 / us / 1e6:5.2f} TB/s", flush=True)
print(f"total {tot / 1e3:.2f} ms per step over these 32 launches")
