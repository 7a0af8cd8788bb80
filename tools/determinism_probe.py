"""Run the same training steps twice from the same state and report which parameter tensors differ bit-wise (GPU box).
usage: determinism_probe.py [config] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsimilar_amd import train as T
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = dict(T.CONFIGS[name])
if len(sys.argv) > 3:
    cfg["batch"] = int(sys.argv[3])
runs = []
for r in range(2):
    model = T.build_model(cfg, "cuda", seed=0)
    ts = T.TrainStep(model, cfg["kind"], 100)
    losses = []
    for i in range(steps):
        l, _ = ts.step(T.synthetic_batch(cfg, "cuda", seed=10 + i))
        losses.append(l.item())
    torch.cuda.synchronize()
    runs.append(({k: v.detach().clone() for k, v in model.state_dict().items()}, losses))
    del model, ts
a, b = runs
print("losses", a[1], b[1], "identical" if a[1] == b[1] else "DIFFER")
bad = {}
for k in a[0]:
    if not torch.equal(a[0][k], b[0][k]):
        d = (a[0][k].float() - b[0][k].float()).abs().max().item()
        kind = k.split(".")[-2] + "." + k.split(".")[-1] if "layer" in k or "blocks" in k else k
        bad.setdefault(kind, []).append(d)
print(f"{sum(len(v) for v in bad.values())} of {len(a[0])} tensors differ")
for k, v in sorted(bad.items(), key=lambda kv: -max(kv[1]))[:40]:
    print(f"  {k:50s} x{len(v):3d}  max |d| {max(v):.3e}")
