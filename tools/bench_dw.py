"""Depthwise-conv forward micro-benchmark across kernel variants (GPU box).  MMSIM_DW_VARIANT = V (0 row-at-a-time, 1 packed up-front, 2 packed 3-row batches)."""
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from multimodalsimilar_amd import ops
from multimodalsimilar_amd._lib import lib
s = ops._stream()
scr = torch.empty(8 << 20, device="cuda")
for (B, H, C, K, S) in ((256, 112, 48, 3, 1), (256, 112, 144, 3, 2), (256, 56, 192, 3, 1), (256, 28, 336, 5, 1), (256, 14, 960, 5, 1), (256, 7, 1632, 5, 1)):
    Ho = H // S
    a = torch.randn(B * H * H, C, device="cuda").bfloat16()
    wT = torch.randn(K * K, C, device="cuda")
    z = torch.empty(B * Ho * Ho, C, dtype=torch.bfloat16, device="cuda")
    sums = torch.zeros(2 * C, device="cuda")
    f = lambda: lib.dwconv_fwd(a.data_ptr(), wT.data_ptr(), z.data_ptr(), sums.data_ptr(), B, H, H, C, K, S, scr.data_ptr(), scr.numel(), s)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    gb = (a.numel() + z.numel()) * 2 / 1e9
    print(f"   {H}x{H}x{C} k{K}s{S}: {t*1e3:7.1f} us  {gb/t*1e3:7.1f} GB/s")
'''
for v in (0, 1, 2):
    print(f"variant {v}", flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MMSIM_DW_VARIANT=str(v)))
