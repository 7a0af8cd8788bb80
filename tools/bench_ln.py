"""add_ln_fwd / ln_bwd micro-benchmark at the text tower's shape [32768, 1024], rotating buffer sets (operands from HBM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
M, H, NSET = 32768, 1024, 6
bf = torch.bfloat16
R = lambda: torch.randn(M, H, device="cuda").to(bf)
t, r = [R() for _ in range(NSET)], [R() for _ in range(NSET)]
y, h = [torch.empty(M, H, dtype=bf, device="cuda") for _ in range(NSET)], [torch.empty(M, H, dtype=bf, device="cuda") for _ in range(NSET)]
g, b = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda")
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
dg, db, dbias = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
def tm(fs, n=5):
    for f in fs: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        for f in fs: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * len(fs)) * 1e3
for p in (0.0, 0.1):
    f = tm([(lambda i=i: ops.add_ln_fwd(t[i], r[i], g, b, y[i], h[i], mean, rstd, 1e-12, p, 1, 3)) for i in range(NSET)])
    bw = tm([(lambda i=i: ops.ln_bwd(t[i], None, y[i], mean, rstd, g, h[i], r[i] if p else None, dg, db, dbias, p, 1, 3)) for i in range(NSET)])
    print(f"dropout {p}: add_ln_fwd {f:6.1f} us ({4 * M * H * 2 / f / 1e6:4.2f} TB/s)   ln_bwd {bw:6.1f} us ({(3 + (1 if p else 0)) * M * H * 2 / bw / 1e6:4.2f} TB/s)")
