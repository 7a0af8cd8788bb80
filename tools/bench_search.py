"""Similarity-search benchmark (GPU box): every vector against the whole set, top-13, as nlp_infer.py:145-152 does."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import search
for N, D in ((20000, 768), (100000, 768), (100000, 2816)):
    x = torch.randn(N, D, device="cuda")
    search.topk_inner_product(x[:512], x, 13)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Dv, I = search.topk_inner_product(x, x, 13)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = bool((I[:, 0] == torch.arange(N, device="cuda")).float().mean() > 0.999)
    print(f"N={N} D={D} k=13: {dt*1e3:8.1f} ms  ({N/dt:,.0f} queries/s, {2.0*N*N*D/dt/1e12:6.1f} TFLOP/s algorithmic fp32-equivalent)  self-first={ok}")
