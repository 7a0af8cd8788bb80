"""Image-tower 1x1-conv GEMM micro-benchmark (GPU box): the six products of each distinct MBConv shape at B=256,
time and effective GB/s (operand + result bytes; these products are HBM-bound)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
from multimodalsimilar_amd._lib import lib
from multimodalsimilar_amd.effnet import build_arch
s = ops._stream()
B = 256
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
seen, tot = set(), {}
h = 112
rows = []
for b in build_arch("efficientnet_b4").blocks:
    ho = h // b.stride
    key = (b.type, b.cin, b.mid, b.cout, h, ho)
    cnt = 1
    if key in seen:
        for r in rows:
            if r[0] == key: r[1] += 1
        h = ho
        continue
    seen.add(key); rows.append([key, 1]); h = ho
bf = torch.bfloat16
R = lambda *sh: torch.randn(*sh, device="cuda").to(bf)
total = 0.0
for key, cnt in rows:
    typ, cin, mid, cout, h, ho = key
    Pi, Po = B * h * h, B * ho * ho
    out = []
    if typ == "ir":
        x, w1 = R(Pi, cin), R(mid, cin); z1 = torch.empty(Pi, mid, dtype=bf, device="cuda")
        out.append(("expand", t(lambda: ops.gemm(x, w1, z1)), (Pi * cin + Pi * mid) * 2))
        dz1 = R(Pi, mid); gw1 = torch.zeros(mid, cin, device="cuda"); dxi = torch.empty(Pi, cin, dtype=bf, device="cuda")
        sk = ops.pick_split_k(mid, cin, Pi)
        out.append(("exp.dW", t(lambda: ops.gemm(dz1, x, gw1, trans_a=True, b_kmajor=False, split_k=sk, accumulate=True)), (Pi * cin + Pi * mid) * 2))
        out.append(("exp.dX", t(lambda: ops.gemm(dz1, w1, dxi, b_kmajor=False)), (Pi * cin + Pi * mid) * 2))
        del x, z1, dz1, dxi
    z2, w3 = R(Po, mid), R(cout, mid); z3 = torch.empty(Po, cout, dtype=bf, device="cuda")
    sc, sh, gate = torch.randn(mid, device="cuda"), torch.randn(mid, device="cuda"), torch.rand(B, mid, device="cuda")
    out.append(("project", t(lambda: lib.gemm_bf16_xf(1, Po, cout, mid, z2.data_ptr(), mid, w3.data_ptr(), mid, z3.data_ptr(), cout, 0,
                                                       sc.data_ptr(), sh.data_ptr(), gate.data_ptr(), ho * ho, 1, 0, s)), (Po * mid + Po * cout) * 2))
    dz3 = R(Po, cout); gw3 = torch.zeros(cout, mid, device="cuda"); da2 = torch.empty(Po, mid, dtype=bf, device="cuda")
    sk = ops.pick_split_k(cout, mid, Po)
    out.append(("prj.dW", t(lambda: lib.gemm_bf16_xf(2, cout, mid, Po, dz3.data_ptr(), cout, z2.data_ptr(), mid, gw3.data_ptr(), mid, 1,
                                                      sc.data_ptr(), sh.data_ptr(), gate.data_ptr(), ho * ho, sk, 1, s)), (Po * mid + Po * cout) * 2))
    out.append(("prj.dX", t(lambda: ops.gemm(dz3, w3, da2, b_kmajor=False)), (Po * mid + Po * cout) * 2))
    del z2, z3, dz3, da2
    line = f"x{cnt} {typ} cin={cin:4d} mid={mid:5d} cout={cout:4d} H={h:3d}->{ho:3d}: "
    for nm, us, by in out:
        line += f"{nm} {us:6.0f}us {by/us/1e3:5.0f}GB/s | "
        total += us * cnt
    print(line, flush=True)
print(f"total {total/1e3:.2f} ms per step (sum over all {sum(c for _, c in rows)} blocks)")
