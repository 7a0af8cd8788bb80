"""The text tower's forward / data-gradient products with their real epilogues: time, TFLOP/s and max error against torch (used to
A/B kernel variants selected by an environment switch or by MMSIM_LIB: run once per setting)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
dev = "cuda"; M = 32768
def bench(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []; tot = 0.0
for name, N, K, epi, kmaj in (("qkv_fwd", 3072, 1024, ops.EPI_NONE, True), ("o_fwd_add", 1024, 1024, ops.EPI_ADD, True),
                              ("ffn1_fwd_pair", 4096, 1024, ops.EPI_GELU_DGELU, True), ("ffn2_fwd_add", 1024, 4096, ops.EPI_ADD, True),
                              ("ffn2_dgrad_mul", 4096, 1024, ops.EPI_MUL, False), ("ffn1_dgrad", 1024, 4096, ops.EPI_NONE, False),
                              ("qkv_dgrad", 1024, 3072, ops.EPI_NONE, False), ("o_dgrad", 1024, 1024, ops.EPI_NONE, False)):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16() if kmaj else (torch.randn(K, N, device=dev) * 0.05).bfloat16()
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    aux = torch.randn(M, N, device=dev).bfloat16()
    aux_o = torch.empty_like(y)
    bias = torch.randn(N, device=dev) * 0.1
    kw = dict(b_kmajor=kmaj, epilogue=epi)
    if kmaj: kw["bias"] = bias
    if epi in (ops.EPI_ADD, ops.EPI_MUL): kw["aux_in"] = aux
    if epi == ops.EPI_GELU_DGELU: kw["aux_out"] = aux_o
    ops.gemm(x, w, y, **kw)
    rows = slice(0, M, 37)                       # every 37th row: all tile rows, both halves of the last tile column
    pre = x[rows].float() @ (w.float().t() if kmaj else w.float())
    if kmaj: pre = pre + bias
    if epi == ops.EPI_ADD: ref = pre + aux[rows].float()
    elif epi == ops.EPI_MUL: ref = pre * aux[rows].float()
    elif epi == ops.EPI_GELU_DGELU: ref = torch.nn.functional.gelu(pre)
    else: ref = pre
    err = ((y[rows].float() - ref).abs().max() / ref.abs().max()).item()
    t = bench(lambda: ops.gemm(x, w, y, **kw)); tot += t
    out.append(f"{name} {t*1e3:.1f}us {2.0*M*N*K/t/1e9:.0f}TF err {err:.1e}")
print(" | ".join(out), f"| sum {tot*1e3:.0f}us", flush=True)
