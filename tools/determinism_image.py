"""Find the first image-tower kernel whose output differs between two identical forward passes (GPU box)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
warnings.simplefilter("ignore")
from multimodalsimilar_amd.effnet import EfficientNet
name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b0"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
res = int(sys.argv[3]) if len(sys.argv) > 3 else 224
m = EfficientNet(name, seed=0).to("cuda").train()
x = torch.randn(B, 3, res, res, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
outs = []
for r in range(2):
    st = m._run_forward(x)
    torch.cuda.synchronize()
    rec = [("z0", st.z0.clone()), ("x0", st.x0.clone())]
    for i, bs in enumerate(st.blocks):
        for k in ("z1", "a1", "z2", "a2", "s", "gate", "z3"):
            t = getattr(bs, k, None)
            if t is not None:
                rec.append((f"b{i}.{k}", t.clone()))
    rec += [("zh", st.zh.clone()), ("pooled", st.pooled.clone()), ("bnstat", st.bnstat.clone())]
    outs.append(rec)
n = 0
for (k, a), (_, b) in zip(*outs):
    if not torch.equal(a, b):
        d = (a.float() - b.float()).abs()
        print(f"{k:12s} differs: max |d| {d.max().item():.3e}  ({(d > 0).float().mean().item() * 100:.3f} % of elements), scale {a.float().abs().max().item():.3e}")
        n += 1
        if n > 12:
            break
print("first divergences listed above" if n else "identical forward passes")
