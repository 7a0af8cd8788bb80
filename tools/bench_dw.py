"""Depthwise-conv micro-benchmark (GPU box): forward / backward-data / backward-weight on the EfficientNet-B4 stride-1
shapes, both forward load forms (MMSIM_DW_VARIANT).  Prints us and effective GB/s."""
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from multimodalsimilar_amd import ops
from multimodalsimilar_amd._lib import lib
s = ops._stream()
scr = torch.empty(8 << 20, device="cuda")
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, C, K) in ((256, 112, 48, 3), (256, 56, 192, 3), (256, 28, 336, 5), (256, 14, 672, 3), (256, 14, 960, 5), (256, 7, 1632, 5)):
    a = torch.randn(B * H * H, C, device="cuda").bfloat16()
    dz = torch.randn(B * H * H, C, device="cuda").bfloat16()
    z1 = torch.randn(B * H * H, C, device="cuda").bfloat16()
    wT = torch.randn(K * K, C, device="cuda")
    z = torch.empty(B * H * H, C, dtype=torch.bfloat16, device="cuda")
    sums = torch.zeros(2 * C, device="cuda")
    v = [torch.randn(C, device="cuda") for _ in range(4)]
    gT = torch.zeros(K * K, C, device="cuda")
    mb = a.numel() * 2 / 1e6
    tf = t(lambda: lib.dwconv_fwd(a.data_ptr(), wT.data_ptr(), z.data_ptr(), sums.data_ptr(), B, H, H, C, K, 1, scr.data_ptr(), scr.numel(), s))
    tb = t(lambda: lib.dwconv_bwd_data(dz.data_ptr(), wT.data_ptr(), z1.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                                       None, z.data_ptr(), sums.data_ptr(), B, H, H, C, K, 1, scr.data_ptr(), scr.numel(), s))
    tw = t(lambda: lib.dwconv_bwd_weight(dz.data_ptr(), a.data_ptr(), gT.data_ptr(), B, H, H, C, K, 1, scr.data_ptr(), scr.numel(), s))
    print(f"   {H}x{H}x{C} k{K}: fwd {tf:7.1f} us {2*mb/tf*1e3:6.0f} GB/s | bwd_data {tb:7.1f} us {3*mb/tb*1e3:6.0f} GB/s | bwd_weight {tw:7.1f} us {2*mb/tw*1e3:6.0f} GB/s")
'''
for v in (0, 1):
    print(f"MMSIM_DW_VARIANT={v} (forward: 0 row at a time, 1 packed up-front loads)", flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MMSIM_DW_VARIANT=str(v)))
