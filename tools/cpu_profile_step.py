"""Host-side cost of one training step (GPU box): cProfile over a few steps of cfg4; the launch path is Python + ctypes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import torch
from multimodalsimilar_amd import train as T
cfg = dict(T.CONFIGS["cfg4"])
model = T.build_model(cfg, "cuda", seed=0, dropout=True)
ts = T.TrainStep(model, cfg["kind"], num_training_steps=1000)
batch = T.synthetic_batch(cfg, "cuda", seed=3)
for _ in range(2):
    ts.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    ts.step(batch)
t_launch = (time.perf_counter() - t0) / 3
torch.cuda.synchronize()
print(f"host time to ENQUEUE one step (no sync): {t_launch*1e3:.1f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    ts.step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
