"""Early-stage projection conv: streaming kernel (mmsim_pw_project_fwd) vs the generic GEMM with statistics, B4 shapes at B=256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
from multimodalsimilar_amd._lib import lib
s = ops._stream(); bf = torch.bfloat16
scr = torch.empty(16 << 20, device="cuda")
def t(f, n=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 256
for name, hw, mid, cout in (("ds0 112^2", 12544, 48, 24), ("ds1 112^2", 12544, 24, 24), ("1.0 56^2", 3136, 144, 32), ("1.x 56^2", 3136, 192, 32),
                            ("2.0 28^2", 784, 192, 56), ("2.x 28^2", 784, 336, 56)):
    P = B * hw
    if not lib.pw_project_fwd_eligible(P, hw, mid, cout):
        print(f"{name}: not eligible (generic GEMM path)"); continue
    a2 = torch.randn(P, mid, device="cuda").to(bf); gate = torch.rand(B, mid, device="cuda"); w3 = torch.randn(cout, mid, device="cuda").to(bf)
    z3 = torch.empty(P, cout, dtype=bf, device="cuda"); sums = torch.zeros(2 * cout, device="cuda")
    tn = t(lambda: lib.pw_project_fwd(a2.data_ptr(), gate.data_ptr(), w3.data_ptr(), z3.data_ptr(), sums.data_ptr(), P, hw, mid, cout, scr.data_ptr(), scr.numel(), s))
    tg = t(lambda: lib.gemm_bf16_bnstats(1, P, cout, mid, a2.data_ptr(), mid, w3.data_ptr(), mid, z3.data_ptr(), cout, None, None, gate.data_ptr(), hw,
                                         sums.data_ptr(), scr.data_ptr(), scr.numel(), s))
    by = (P * mid + P * cout) * 2
    line = f"{name:10s} mid={mid:3d} cout={cout:2d}: fwd streaming {tn:6.0f} us {by/tn/1e6:5.2f} TB/s | gemm {tg:6.0f} us {by/tg/1e6:5.2f} TB/s"
    if lib.pw_project_bwd_eligible(P, hw, mid, cout):
        dz3 = torch.randn(P, cout, device="cuda").to(bf); da = torch.empty(P, mid, dtype=bf, device="cuda"); dw = torch.zeros(cout, mid, device="cuda")
        sk = ops.pick_split_k(cout, mid, P)
        tb = t(lambda: lib.pw_project_bwd(dz3.data_ptr(), a2.data_ptr(), gate.data_ptr(), w3.data_ptr(), da.data_ptr(), dw.data_ptr(), P, hw, mid, cout, scr.data_ptr(), scr.numel(), s))
        def pair():
            lib.gemm_group_begin()
            lib.gemm_bf16_xf(2, cout, mid, P, dz3.data_ptr(), cout, a2.data_ptr(), mid, dw.data_ptr(), mid, 1, None, None, gate.data_ptr(), hw, sk, 1, s)
            ops.gemm(dz3, w3, da, b_kmajor=False)
            lib.gemm_group_end()
        tp = t(pair)
        byb = (2 * P * mid + 2 * P * cout) * 2
        line += f" || bwd streaming {tb:6.0f} us {byb/tb/1e6:5.2f} TB/s | gemm pair {tp:6.0f} us {byb/tp/1e6:5.2f} TB/s"
    print(line, flush=True)
for name, hw, cin, mid in (("1.0 112^2 expand", 12544, 24, 144), ("1.x 56^2 expand", 3136, 32, 192)):
    P = B * hw
    x = torch.randn(P, cin, device="cuda").to(bf); w1 = torch.randn(mid, cin, device="cuda").to(bf)
    z1 = torch.empty(P, mid, dtype=bf, device="cuda"); sums = torch.zeros(2 * mid, device="cuda")
    tn = t(lambda: lib.pw_expand_fwd(x.data_ptr(), w1.data_ptr(), z1.data_ptr(), sums.data_ptr(), P, mid, cin, scr.data_ptr(), scr.numel(), s))
    tg = t(lambda: lib.gemm_bf16_bnstats(0, P, mid, cin, x.data_ptr(), cin, w1.data_ptr(), cin, z1.data_ptr(), mid, None, None, None, 1,
                                         sums.data_ptr(), scr.data_ptr(), scr.numel(), s))
    by = (P * mid + P * cin) * 2
    print(f"{name:18s} cin={cin} mid={mid}: streaming {tn:6.0f} us {by/tn/1e6:5.2f} TB/s | gemm {tg:6.0f} us {by/tg/1e6:5.2f} TB/s", flush=True)
