"""mmsim_pw_expand_bwd on the 56^2 / 28^2 shapes of B4 at B=256: time and effective TB/s (reads d(pre1), z1, x; writes dx)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
from multimodalsimilar_amd._lib import lib
s = ops._stream(); bf = torch.bfloat16
scr = torch.empty(32 << 20, device="cuda")
def t(f, n=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 256
for name, hw, cin, mid in (("1.x 56^2", 3136, 32, 192), ("2.x 28^2", 784, 56, 336)):
    P = B * hw
    dpre = torch.randn(P, mid, device="cuda").to(bf); z1 = torch.randn(P, mid, device="cuda").to(bf); x = torch.randn(P, cin, device="cuda").to(bf)
    w1 = torch.randn(mid, cin, device="cuda").to(bf); dx = torch.empty(P, cin, dtype=bf, device="cuda"); dw = torch.zeros(mid, cin, device="cuda")
    v = [torch.rand(mid, device="cuda") + 0.5 for _ in range(3)]; sums = torch.randn(2 * mid, device="cuda"); dg = torch.zeros(mid, device="cuda"); db = torch.zeros(mid, device="cuda")
    tt = t(lambda: lib.pw_expand_bwd(dpre.data_ptr(), z1.data_ptr(), x.data_ptr(), None, w1.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(),
                                     sums.data_ptr(), dx.data_ptr(), dw.data_ptr(), dg.data_ptr(), db.data_ptr(), P, mid, cin, scr.data_ptr(), scr.numel(), s))
    by = (2 * P * mid + 2 * P * cin) * 2
    res = torch.randn(P, cin, device="cuda").to(bf)
    tr = t(lambda: lib.pw_expand_bwd(dpre.data_ptr(), z1.data_ptr(), x.data_ptr(), res.data_ptr(), w1.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(),
                                     sums.data_ptr(), dx.data_ptr(), dw.data_ptr(), dg.data_ptr(), db.data_ptr(), P, mid, cin, scr.data_ptr(), scr.numel(), s))
    print(f"{name} cin={cin} mid={mid}: {tt:6.0f} us {by/tt/1e6:5.2f} TB/s | with residual rows {tr:6.0f} us {(by + P * cin * 2)/tr/1e6:5.2f} TB/s", flush=True)
