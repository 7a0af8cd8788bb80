"""What would a DMA-pipelined GEMM give on the late-stage 1x1-conv shapes?  Generic kernel on the true (ragged) shape vs the
fast LDS-DMA kernels on the same shape padded up to tile multiples (M%256, N%128, K%64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
bf = torch.bfloat16
R = lambda *sh: torch.randn(*sh, device="cuda").to(bf)
def t(f, n=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
up = lambda v, m: -(-v // m) * m
def three(P, cin, mid, pad):
    if pad: P, cin_n, cin_k, mid_n, mid_k = up(P, 256), up(cin, 128), up(cin, 64), up(mid, 128), up(mid, 64)
    else: cin_n = cin_k = cin; mid_n = mid_k = mid
    res = []
    x, w = R(P, cin_k), R(mid_n, cin_k); z = torch.empty(P, mid_n, dtype=bf, device="cuda")
    res.append(t(lambda: ops.gemm(x, w, z)))                                     # expand fwd  [P,cin]x[mid,cin]^T
    dz, w2 = R(P, mid_k), R(mid_k, cin_n); dx = torch.empty(P, cin_n, dtype=bf, device="cuda")
    res.append(t(lambda: ops.gemm(dz, w2, dx, b_kmajor=False)))                  # expand dgrad [P,mid]x[mid,cin]
    dzm, xm = R(P, up(mid, 256) if pad else mid), R(P, cin_n); gw = torch.zeros(dzm.shape[1], cin_n, device="cuda")
    sk = max(1, min(64, 512 // max(1, (gw.shape[0] // 256 if pad else -(-mid // 128)) * max(1, cin_n // 128))))
    res.append(t(lambda: ops.gemm(dzm, xm, gw, trans_a=True, b_kmajor=False, split_k=sk, accumulate=True)))   # expand wgrad
    dy, w3 = R(P, cin_k), R(cin_k, mid_n); da = torch.empty(P, mid_n, dtype=bf, device="cuda")
    res.append(t(lambda: ops.gemm(dy, w3, da, b_kmajor=False)))                  # project dgrad [P,cout]x[cout,mid]
    return res
for name, P, cin, mid in (("s3 14^2", 50176, 112, 672), ("s4 14^2", 50176, 160, 960), ("s5 7^2", 12544, 272, 1632), ("s6 7^2", 12544, 448, 2688),
                          ("head 7^2", 12544, 448, 1792)):
    a = three(P, cin, mid, False); b = three(P, cin, mid, True)
    print(f"{name:9s} cin={cin} mid={mid}: generic fwd {a[0]:5.0f} dX {a[1]:5.0f} dW {a[2]:5.0f} prj.dX {a[3]:5.0f} us | padded+DMA fwd {b[0]:5.0f} dX {b[1]:5.0f} dW {b[2]:5.0f} prj.dX {b[3]:5.0f} us", flush=True)
