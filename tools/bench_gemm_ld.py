"""Leading-dimension sensitivity of the pipelined GEMM on the text tower's shapes (M = 32768): activations with power-of-two row strides
(2048 / 8192 bytes) against the same tensors padded by 64 / 128 elements.  nt: y = x W^T; nn: dx = dy W; tn: dW = dy^T x (split-K)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodalsimilar_amd import ops
M = 32768
PAD = int(os.environ.get("LD_PAD", "64"))
def view(rows, cols, pad, dt=torch.bfloat16, rnd=True):
    buf = (torch.randn(rows, cols + pad, device="cuda") * 0.05).to(dt) if rnd else torch.zeros(rows, cols + pad, device="cuda", dtype=dt)
    return buf[:, :cols]
def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = {0: 0.0, 1: 0.0}
for name, N, K in (("qkv", 3072, 1024), ("o", 1024, 1024), ("ffn1", 4096, 1024), ("ffn2", 1024, 4096)):
    fl = 2.0 * M * N * K
    row = []
    for pad in (0, PAD):
        x = view(M, K, pad); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); y = view(M, N, pad)
        wt = w.t().contiguous()          # [K, N] for the nn layout... dx = dy @ w uses w [N, K] stored [K? no: b_kmajor=False means B is [K][N]
        dy = view(M, N, pad); dx = view(M, K, pad)
        dw = torch.zeros(N, K, device="cuda")
        sk = ops.pick_split_k(N, K, M)
        t_nt = t(lambda: ops.gemm(x, w, y))
        t_nn = t(lambda: ops.gemm(dy, w, dx, b_kmajor=False))
        t_tn = t(lambda: ops.gemm(dy, x, dw, trans_a=True, b_kmajor=False, split_k=sk, accumulate=True))
        tot[1 if pad else 0] += t_nt + t_nn + t_tn
        row.append(f"pad {pad:3d}: NT {fl/t_nt/1e9:6.0f} NN {fl/t_nn/1e9:6.0f} TN {fl/t_tn/1e9:6.0f} TF")
    print(f"{name:5s} " + " | ".join(row), flush=True)
print(f"layer total: unpadded {tot[0]:.3f} ms, padded {tot[1]:.3f} ms")
