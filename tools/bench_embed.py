"""mmsim_embed_ln_bwd at the text tower's shape (B=256, S=128, H=1024, vocab 21128): time, with ids as the bench draws them
(uniform, [CLS] at position 0) and with all-distinct ids (no collisions in the word-gradient scatter)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import ops
B, S, H, V = 256, 128, 1024, 21128
dev = "cuda"
word = torch.randn(V, H, device=dev) * 0.02; pos = torch.randn(512, H, device=dev) * 0.02; typ = torch.randn(2, H, device=dev) * 0.02
gamma = torch.ones(H, device=dev)
dout = torch.randn(B * S, H, device=dev).bfloat16()
tts = torch.zeros(B, S, dtype=torch.int64, device=dev)
err = torch.zeros(4, dtype=torch.int32, device=dev)
dword = torch.zeros(V, H, device=dev); dpos = torch.zeros(512, H, device=dev); dtyp = torch.zeros(2, H, device=dev)
dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, ids in (("uniform ids, [CLS] first", torch.randint(0, V, (B, S), device=dev)), ("distinct ids", (torch.arange(B * S, device=dev) % V).view(B, S))):
    if name.startswith("uniform"): ids[:, 0] = 101
    for p in (0.0, 0.1):
        us = t(lambda: ops.embed_ln_bwd(dout, ids, tts, word, pos, typ, gamma, dword, dpos, dtyp, dg, db, B, S, H, 1e-12, err, dropout_p=p, seed=1, stream_id=3))
        print(f"{name:26s} dropout {p}: {us:7.1f} us", flush=True)
