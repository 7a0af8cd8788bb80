"""Stem-conv micro-benchmark (GPU box): forward and weight gradient at the cfg4 shape (B=256, 224^2, 48 channels),
time and effective GB/s over the algorithmic bytes (image read once + z / dz once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from multimodalsimilar_amd import ops
from multimodalsimilar_amd._lib import lib
s = ops._stream()
B, H, Co = 256, 224, int(os.environ.get("CO", 48))
x = torch.randn(B, 3, H, H, device="cuda")
w = torch.randn(Co, 3, 3, 3, device="cuda") * 0.2
Ho = H // 2
z = torch.empty(B * Ho * Ho, Co, dtype=torch.bfloat16, device="cuda")
sums = torch.zeros(2 * Co, device="cuda")
scr = torch.empty(8 << 20, device="cuda")
dz = torch.randn(B * Ho * Ho, Co, device="cuda").bfloat16()
dw = torch.zeros_like(w)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fwd = lambda: lib.stem_fwd(x.data_ptr(), w.data_ptr(), z.data_ptr(), sums.data_ptr(), B, H, H, Co, scr.data_ptr(), scr.numel(), s)
wg = lambda: lib.stem_wgrad(dz.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, H, Co, scr.data_ptr(), scr.numel(), s)
by = x.numel() * 4 + z.numel() * 2
tf, tw = t(fwd), t(wg)
print(f"stem fwd {tf:7.1f} us {by/tf/1e3:6.0f} GB/s   wgrad {tw:7.1f} us {by/tw/1e3:6.0f} GB/s")
# correctness at full size against torch (bf16 rounding of the stored output only)
sums.zero_(); fwd()
ref = F.conv2d(x, w, None, stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
print("fwd relerr", ((z.float() - ref).norm() / ref.norm()).item(), " sums relerr",
      ((sums[:Co] - z.float().sum(0)).norm() / z.float().sum(0).norm()).item(),
      ((sums[Co:] - (z.float() ** 2).sum(0)).norm() / (z.float() ** 2).sum(0).norm()).item())
dw.zero_(); wg()
xr = x[:32]; wr = w.clone().requires_grad_(True)
dw32 = torch.zeros_like(w)
lib.stem_wgrad(dz.data_ptr(), xr.data_ptr(), dw32.data_ptr(), 32, H, H, Co, scr.data_ptr(), scr.numel(), s)
F.conv2d(xr, wr, None, stride=2, padding=1).backward(dz[:32 * Ho * Ho].float().view(32, Ho, Ho, Co).permute(0, 3, 1, 2))
print("wgrad relerr (32 images)", ((dw32 - wr.grad).norm() / wr.grad.norm()).item())
