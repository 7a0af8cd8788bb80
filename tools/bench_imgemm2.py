"""Image-tower 1x1-conv products on the GENERIC GEMM kernel at the late-stage shapes of EfficientNet-B4 @ 224, B = 256 (GPU box).
Forward tensors fp16, gradients bf16 (include/mmsim_hip.h, EfficientNet section).  Every product walks NSET rotating operand sets
(> 256 MB in total) so that operands come from HBM as in the step, not from the Infinity Cache a repeated buffer would sit in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
from multimodalsimilar_amd._lib import lib
s = ops._stream()
B, NSET = 256, 4
H16, BF = torch.float16, torch.bfloat16
R = lambda dt, *sh: torch.randn(*sh, device="cuda").to(dt)
def t(fs, n=3):
    for f in fs: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        for f in fs: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * len(fs)) * 1e3
# (count, H, cin, mid, cout)
shapes = [(5, 14, 112, 672, 112), (1, 14, 112, 672, 160), (5, 14, 160, 960, 160), (7, 7, 272, 1632, 272), (1, 7, 272, 1632, 448), (1, 7, 448, 2688, 448)]
total = 0.0
for cnt, H, cin, mid, cout in shapes:
    P, HW = B * H * H, H * H
    x = [R(H16, P, cin) for _ in range(NSET)]; w1 = R(H16, mid, cin); w1b = w1.float().to(BF)
    z1 = [torch.empty(P, mid, dtype=H16, device="cuda") for _ in range(NSET)]
    a2 = [R(H16, P, mid) for _ in range(NSET)]; w3 = R(H16, cout, mid); w3b = w3.float().to(BF)
    z3 = [torch.empty(P, cout, dtype=H16, device="cuda") for _ in range(NSET)]
    gate = torch.rand(B, mid, device="cuda")
    sums1, sums3 = torch.zeros(2 * mid, device="cuda"), torch.zeros(2 * cout, device="cuda")
    scr = torch.empty(8 << 20, device="cuda")
    dz3 = [R(BF, P, cout) for _ in range(NSET)]; da = [torch.empty(P, mid, dtype=BF, device="cuda") for _ in range(NSET)]
    dz1 = [R(BF, P, mid) for _ in range(NSET)]; dx = [torch.empty(P, cin, dtype=BF, device="cuda") for _ in range(NSET)]
    res = R(BF, P, cin)
    gw1, gw3 = torch.zeros(mid, cin, device="cuda"), torch.zeros(cout, mid, device="cuda")
    sk1, sk3 = ops.pick_split_k(mid, cin, P), ops.pick_split_k(cout, mid, P)
    _d = int(os.environ.get('SK_DIV', '1')); sk1, sk3 = max(1, sk1 // _d), max(1, sk3 // _d)
    def f_exp(i): return lambda: lib.gemm_bf16_bnstats(0, P, mid, cin, x[i].data_ptr(), cin, w1.data_ptr(), cin, z1[i].data_ptr(), mid, None, None, None, 1, sums1.data_ptr(), scr.data_ptr(), scr.numel(), s)
    def f_prj(i): return lambda: lib.gemm_bf16_bnstats(1, P, cout, mid, a2[i].data_ptr(), mid, w3.data_ptr(), mid, z3[i].data_ptr(), cout, None, None, gate.data_ptr(), HW, sums3.data_ptr(), scr.data_ptr(), scr.numel(), s)
    def f_bprj(i):
        def g():
            with ops.gemm_group():
                lib.gemm_bf16_xf(2, cout, mid, P, dz3[i].data_ptr(), cout, a2[i].data_ptr(), mid, gw3.data_ptr(), mid, 1, None, None, gate.data_ptr(), HW, sk3, 1, s)
                ops.gemm(dz3[i], w3b, da[i], b_kmajor=False)
        return g
    def f_bexp(i):
        def g():
            with ops.gemm_group():
                ops.gemm(dz1[i], x[i], gw1, trans_a=True, b_kmajor=False, split_k=sk1, accumulate=True)
                ops.gemm(dz1[i], w1b, dx[i], b_kmajor=False, epilogue=ops.EPI_ADD, aux_in=res)
        return g
    if os.environ.get("SPLIT_PAIRS"):      # the two members of each backward pair on their own (no paired launch)
        def f_w3(i): return lambda: lib.gemm_bf16_xf(2, cout, mid, P, dz3[i].data_ptr(), cout, a2[i].data_ptr(), mid, gw3.data_ptr(), mid, 1, None, None, gate.data_ptr(), HW, sk3, 1, s)
        def f_d3(i): return lambda: ops.gemm(dz3[i], w3b, da[i], b_kmajor=False)
        def f_w1(i): return lambda: ops.gemm(dz1[i], x[i], gw1, trans_a=True, b_kmajor=False, split_k=sk1, accumulate=True)
        def f_d1(i): return lambda: ops.gemm(dz1[i], w1b, dx[i], b_kmajor=False, epilogue=ops.EPI_ADD, aux_in=res)
        print(f"   {H:2d}^2 {cin}->{mid}->{cout}: wgrad-project(sk={sk3}) {t([f_w3(i) for i in range(NSET)]):6.1f} | dgrad-project {t([f_d3(i) for i in range(NSET)]):6.1f} | "
              f"wgrad-expand(sk={sk1}) {t([f_w1(i) for i in range(NSET)]):6.1f} | dgrad-expand {t([f_d1(i) for i in range(NSET)]):6.1f} us", flush=True)
    row = []
    for nm, mk, by in (("expand", f_exp, (P * cin + P * mid) * 2), ("project", f_prj, (P * mid + P * cout) * 2),
                       ("bwd-project", f_bprj, (2 * P * cout + 2 * P * mid) * 2), ("bwd-expand", f_bexp, (2 * P * mid + 3 * P * cin) * 2)):
        us = t([mk(i) for i in range(NSET)])
        row.append(f"{nm} {us:6.1f} us {by / us / 1e6:5.2f} TB/s")
        total += us * cnt
    print(f"x{cnt} {H:2d}^2 {cin:4d}->{mid:5d}->{cout:4d}: " + " | ".join(row), flush=True)
    del x, z1, a2, z3, dz3, da, dz1, dx
print(f"total {total / 1e3:.2f} ms per step over these {sum(c for c, *_ in shapes)} blocks")
