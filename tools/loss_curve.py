"""Sanity run on the GPU box: N steps of cfg4 on one fixed synthetic batch; prints the loss every 10 steps (it must fall
monotonically towards zero as the model memorises the batch) and checks for non-finite values."""
import math, os, sys
sys.path.insert(0, os.getcwd())
import torch
from multimodalsimilar_amd import train as T
cfg = dict(T.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
model = T.build_model(cfg, "cuda", seed=0, dropout=True)
ts = T.TrainStep(model, cfg["kind"], num_training_steps=n)
batch = T.synthetic_batch(cfg, "cuda", seed=1234)
for i in range(n):
    loss, pred = ts.step(batch)
    if i % 10 == 0 or i == n - 1:
        l = float(loss.item())
        acc = float((pred == batch["labels"]).float().mean().item())
        print(f"step {i:3d}  loss {l:8.4f}  train-acc {acc:.3f}", flush=True)
        assert math.isfinite(l)
