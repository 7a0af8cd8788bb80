import os, sys, torch
sys.path.insert(0, os.getcwd())
from multimodalsimilar_amd import train as T
import multimodal_classifier as mc
def run(two, cfgname, steps=4, dropout=False):
    mc._TWO_STREAMS = two
    cfg = dict(T.CONFIGS[cfgname])
    if cfgname == "cfg4": cfg["batch"] = 32
    model = T.build_model(cfg, "cuda", seed=0, dropout=dropout)
    ts = T.TrainStep(model, cfg["kind"], num_training_steps=100)
    batch = T.synthetic_batch(cfg, "cuda", seed=3)
    out = []
    for _ in range(steps):
        loss, _ = ts.step(batch)
        out.append(float(loss.item()))
    return out
for name in ("tiny", "cfg4"):
    a = run(False, name); b = run(True, name)
    print(name, "one stream ", [round(x, 4) for x in a])
    print(name, "two streams", [round(x, 4) for x in b])
