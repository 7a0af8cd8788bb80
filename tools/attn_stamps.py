"""Segment timing of attn_bwd_kernel<4> from s_memtime stamps (diagnostic build: tools/build_variant.sh ... attention.hip -DATTN_STAMP=1,
run with MMSIM_LIB pointing at it).  Prints the median cycles of wave 0 per segment over the sampled workgroups."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodalsimilar_amd import ops
from multimodalsimilar_amd._lib import lib, LIBPATH
B, S, nh = 256, 128, 16
H = nh * 64
qkv = torch.randn(B * S, 3 * H, device="cuda").bfloat16()
ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device="cuda")
dctx = torch.randn(B * S, H, device="cuda").bfloat16()
lse = torch.empty(B * nh * S, device="cuda")
dqkv = torch.empty_like(qkv)
db = torch.zeros(3 * H, device="cuda")
ops.attn_fwd(qkv, None, ctx, lse, B, S, nh, H, 0.1, 1, 2)
for _ in range(3):
    ops.attn_bwd(qkv, None, ctx, dctx, lse, dqkv, B, S, nh, H, 0.1, 1, 2, dbias=db)
torch.cuda.synchronize()
dll = ctypes.CDLL(LIBPATH)
buf = (ctypes.c_uint64 * (64 * 64))()
rc = dll.mmsim_debug_attn_stamps(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
names = {0: "start", 1: "prologue loads issued", 2: "tile 0 parked (loads landed)", 3: "barrier", 36: "after loop barrier", 37: "dK/dV staged + barrier", 38: "dK/dV stored", 39: "end"}
for qt in range(4):
    for k, n in ((4, "iter start"), (5, "S / dP MFMAs"), (6, "softmax / dropout"), (7, "dV / dK MFMAs"), (8, "dS^T -> LDS"), (9, "park next tile"), (10, "barrier"), (11, "dQ + store")):
        names[k + 8 * qt] = f"q-tile {qt}: {n}"
order = sorted(names)
rows = a[(a[:, 0] > 0) & (a[:, 39] > a[:, 0])]
print(f"{len(rows)} sampled workgroups; total median {np.median(rows[:, 39] - rows[:, 0]):.0f} cycles (s_memtime ticks)")
prev = order[0]
for k in order[1:]:
    d = rows[:, k] - rows[:, prev]
    print(f"  {names[k]:34s} {np.median(d):8.0f}   (p10 {np.percentile(d, 10):6.0f}, p90 {np.percentile(d, 90):6.0f})")
    prev = k
