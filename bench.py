#!/usr/bin/env python
"""Benchmark of the hot path: the data-parallel two-tower (RoBERTa-wwm-ext-large + EfficientNet-B4) + ArcFace
training step, BASELINE.json's metric, on synthetic inputs resident in HBM.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward (both towers, normalise+concat, fused ArcFace margin + cross-entropy) + backward + gradient
all-reduce (N > 1) + both fused AdamW updates + LR schedules: nothing is skipped or cached inside the timed region.
Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline     - the dominant kernel (the bf16 MFMA GEMM gemm_pp64_kernel<false,true,256>, Y = X W^T): algorithmic
                 FLOPs (2 M N K per launch) / its mean launch duration measured with HIP events on the launch stream
                 over the timed region, against the 2.5 PFLOP/s dense bf16 MFMA peak (MI355X_MICROARCH.md).  The towers
                 run on two HIP streams, so inside the timed region these launches time-share the chip with image-tower
                 kernels; `achieved_exclusive` is the same measurement over two untimed steps with one stream, taken right
                 after the timed region; `traffic` is the fabric-side bytes per launch from separate rocprofv3 --pmc passes
                 (profiles/r04_pmc_gemm_pp64.json);
  cpu_baseline - the oracle's CPU restatement of the same step (oracle/step_ref.py, kind "port") on the host cores,
                 on a bounded sample (same model, B_cpu pairs per step).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
HBM_PEAK_GBS = 8000.0


def algorithmic_flops_per_pair(cfg):
    """fwd+bwd = 3 x fwd, 1 MAC = 2 FLOP; elementwise / softmax / LN not counted."""
    f = 0.0
    if cfg["kind"] in ("nlp", "multimodal"):
        L, H, S = (24, 1024, cfg["seq_len"]) if cfg["text"] == "large" else ((12, 768, cfg["seq_len"]) if cfg["text"] == "base" else (2, 128, cfg["seq_len"]))
        f += 3 * L * (24 * S * H * H + 4 * S * S * H)
        D = H
    if cfg["kind"] in ("cv", "multimodal"):
        from multimodalsimilar_amd.effnet import count_macs
        macs = count_macs(cfg["image"], cfg["res"])
        f += 3 * 2 * macs
        D = (cfg.get("fc_dim", 0) if cfg.get("use_fc") else {"efficientnet_b0": 1280, "efficientnet_b4": 1792}[cfg["image"]]) + \
            (D if cfg["kind"] == "multimodal" else 0)
    f += 6 * D * cfg["classes"]
    return f


def _goes_to_pp64_nt(a, b, c, trans_a, b_kmajor, kw):
    """Mirror of the dispatch in csrc/gemm_fast.hip: forward-layout launches that run gemm_pp64_kernel<false,true,256>."""
    if trans_a or not b_kmajor or kw.get("split_k", 1) != 1:
        return False
    M, N, K = c.shape[0], c.shape[1], a.shape[1]
    return M % 256 == 0 and N % 256 == 0 and K % 64 == 0 and (M // 256) * (N // 256) >= 128 and c.stride(0) % 4 == 0


def _pmc_traffic(config):
    """HBM-side bytes per launch of the dominant kernel (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), measured in separate
    rocprofv3 --pmc passes over the same four forward shapes (tools/pmc_gemm.py) and committed under profiles/; counters
    cannot be read inside this process, so the figure is only reported for the workload it was measured on (cfg4 / cfg3)."""
    path = os.path.join(ROOT, "profiles", "r04_pmc_gemm_pp64.json")
    if config not in ("cfg4", "cfg3") or not os.path.exists(path):
        return None, None
    with open(path) as f:
        d = json.load(f)
    return d["mean_hbm_bytes_per_launch"], sum(r["algorithmic_bytes"] for r in d["per_shape"]) / len(d["per_shape"])


class GemmTimer:
    """HIP events around every launch of the dominant kernel, gemm_pp64_kernel<false,true,256> (Y = X W^T), on the
    launch stream (torch's current stream is the stream the C ABI launches on)."""

    def __init__(self, ops):
        self.ops, self.orig, self.rec, self.on = ops, ops.gemm, [], False

    def install(self):
        def timed(a, b, c, *, trans_a=False, b_kmajor=True, **kw):
            if not self.on or not _goes_to_pp64_nt(a, b, c, trans_a, b_kmajor, kw):
                return self.orig(a, b, c, trans_a=trans_a, b_kmajor=b_kmajor, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.orig(a, b, c, trans_a=trans_a, b_kmajor=b_kmajor, **kw)
            e1.record()
            self.rec.append((e0, e1, 2.0 * c.shape[0] * c.shape[1] * a.shape[1]))
            return r
        self.ops.gemm = timed

    def summary(self):
        if not self.rec:
            return None
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in self.rec)
        fl = sum(f for _, _, f in self.rec)
        return dict(launches=len(self.rec), avg_us=1e3 * ms / len(self.rec), tflops=fl / (ms * 1e-3) / 1e12)


def cpu_baseline(cfg, b_cpu, steps):
    """Oracle (CPU restatement) of the same step on the host cores; bounded sample: b_cpu pairs per step."""
    from oracle import bert_ref, effnet_ref, step_ref
    torch.manual_seed(0)
    # the GPU box gives one GPU a 16-core CPU share; os.cpu_count() reports the whole host (oversubscription)
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    ncores = int(os.environ.get("MMSIM_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(ncores)
    kind = cfg["kind"]
    ts = ti = None
    shape = None
    D = 0
    if kind in ("nlp", "multimodal"):
        big = cfg["text"] == "large"
        shape = bert_ref.BertShape(21128, 1024 if big else 768, 24 if big else 12, 16 if big else 12, 4096 if big else 3072)
        ts = bert_ref.init_state(shape, seed=0)
        D += shape.hidden_size
    if kind in ("cv", "multimodal"):
        ti = effnet_ref.init_state(cfg["image"], fc_dim=cfg.get("fc_dim") if cfg.get("use_fc") else None, seed=0)
        D += cfg["fc_dim"] if cfg.get("use_fc") else effnet_ref.arch(cfg["image"])["head"]
    head = torch.empty(cfg["classes"], D)
    torch.nn.init.xavier_uniform_(head)
    orc = step_ref.TwoTowerOracle(shape, ts, cfg.get("image"), ti, head, num_steps=10 ** 6, use_fc=bool(cfg.get("use_fc")),
                                  margin={"multimodal": 0.5, "nlp": 0.4, "cv": 0.2}[kind])
    from multimodalsimilar_amd.train import synthetic_batch
    batch = synthetic_batch(cfg, "cpu", seed=4321, batch=b_cpu)
    tw = time.perf_counter()
    orc.step(batch)                             # one un-timed warm step (allocator, thread pool, first-touch of the weights)
    warm = time.perf_counter() - tw
    t0 = time.perf_counter()
    done = 0
    # >= 2 timed steps; more only while the sample stays bounded (~30 s of CPU work in all)
    while done < 2 or (done < steps and time.perf_counter() - t0 + warm < 30.0):
        orc.step(batch)
        done += 1
    steps = done
    dt = (time.perf_counter() - t0) / steps
    model = ""
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return dict(value=b_cpu / dt, unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{steps} timed steps (after 1 un-timed warm step) of the identical step at B_cpu={b_cpu} pairs (fp32, torch CPU ops "
                       f"via oracle/step_ref.py), {dt:.2f} s/step; {ncores} threads on: {model}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override (reported; invalid as the headline)")
    ap.add_argument("--cpu-batch", type=int, default=8, help="B_cpu of the cpu_baseline leg (SURVEY 8d: 8 or 16)")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--ragged-masks", action="store_true", help="per-row sequence lengths ~ U{8..S} (SURVEY 8d realism run; not the headline)")
    ap.add_argument("--literal-loss", action="store_true", help="materialised logits + nn.CrossEntropyLoss instead of the fused head")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python (eager) instead of replaying the captured hipGraph")
    ap.add_argument("--shard-head-from", type=int, default=None, help="classes from which an N > 1 run shards the ArcFace head over ranks (default 500000; 100000 shards cfg4's)")
    ap.add_argument("--grad-dtype", choices=("fp32", "bf16"), default=None, help="element type of the gradient all-reduce buckets (default fp32)")
    args = ap.parse_args()
    if args.shard_head_from is not None:
        os.environ["MMSIM_SHARD_HEAD_FROM"] = str(args.shard_head_from)       # read by multimodalsimilar_amd.train at import
    if args.grad_dtype is not None:
        os.environ["MMSIM_GRAD_DTYPE"] = args.grad_dtype                      # read by GradientExchange

    # before the first HIP call of the process: the host driver only supports dmabuf IPC (RCCL / cross-process tensors)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if os.environ.get("MMSIM_SINGLE_DEVICE") == "1":      # rehearsal of the N > 1 path on a one-GPU box (with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("MMSIM_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from multimodalsimilar_amd import build, ops
    if rank == 0:
        build.build(verbose=False)
    if world > 1:
        dist.barrier()
    from multimodalsimilar_amd import train

    cfg = dict(train.CONFIGS[args.config])
    if args.batch:
        cfg["batch"] = args.batch
    model = train.build_model(cfg, device, seed=0, dropout=not args.no_dropout)
    step = train.TrainStep(model, cfg["kind"], num_training_steps=10 ** 6, fused_loss=not args.literal_loss)
    batch = train.synthetic_batch(cfg, device, seed=1234 + rank, ragged_masks=args.ragged_masks)
    timer = GemmTimer(ops)
    timer.install()
    if os.environ.get("MMSIM_MAIN_PRIORITY"):      # experiment: the caller's (text tower's) stream at another HIP priority
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=int(os.environ["MMSIM_MAIN_PRIORITY"])))

    # The step is ~2 500 launches; issued from Python they cost as much host time as the step takes on the GPU.  Single-process
    # runs therefore replay the step as ONE captured hipGraph (train.GraphedTrainStep: same kernels, same order, same streams;
    # step-dependent scalars read from device memory).  Data-parallel runs keep the eager step (collectives are not captured).
    gstep, launch = None, "eager (one Python launch per kernel)"
    if world == 1 and not args.no_graph and os.environ.get("MMSIM_GRAPH", "1") != "0" and not args.literal_loss:
        try:
            gstep = train.GraphedTrainStep(step, batch, warmup=2)
            launch = "hipGraph replay (the whole step captured once: forward, loss, backward, both AdamW updates)"
        except Exception as e:      # never lose the measurement to a capture problem: fall back to the eager step, and say so
            print(f"bench: hipGraph capture failed ({type(e).__name__}: {e}); running the eager step", file=sys.stderr, flush=True)
            step.opt_emb.dev_hyper = step.opt_fc.dev_hyper = None
            ops.lib.set_step_seed_ptr(None)
            gstep = None
    run = gstep.step if gstep is not None else step.step

    for _ in range(args.warmup):
        loss, _ = run(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.on = gstep is None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, pred = run(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.on = False
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    model.classifier.check_labels()
    lossv = float(loss.item())

    # roofline of the dominant kernel: HIP events around each of its launches.  Events cannot bracket a launch inside a replayed
    # graph, so with the graph path they are taken over the same number of EAGER steps right after the timed region (the graph
    # replays exactly these kernels; the rocprofv3 summary under profiles/ is of the replayed graph itself).
    roof_from = "the timed region"
    eager_ms = None
    if gstep is not None:
        gstep.close()
        # the EAGER step (one Python launch per kernel) is what every data-parallel run executes (collectives are not captured):
        # timed here in the same process, on the same box, right after the graph replay, and reported beside it
        step.step(batch)
        torch.cuda.synchronize()
        te = time.perf_counter()
        for _ in range(min(args.steps, 5)):
            step.step(batch)
        torch.cuda.synchronize()
        eager_ms = 1e3 * (time.perf_counter() - te) / min(args.steps, 5)
        timer.on = True
        for _ in range(min(args.steps, 5)):
            step.step(batch)
        torch.cuda.synchronize()
        timer.on = False
        roof_from = f"{min(args.steps, 5)} eager steps right after the timed region (same kernels as the replayed graph)"

    # The towers overlap on two HIP streams inside the timed region, so the dominant kernel's launches above share the CUs
    # with image-tower kernels.  Its rate with the chip to itself is measured right after, outside the timed region
    # (2 untimed steps with both towers on one stream); reported as roofline.achieved_exclusive, never as `achieved`.
    g_excl = None
    if cfg["kind"] == "multimodal" and rank == 0 and world == 1:
        import multimodal_classifier as _mc
        if _mc._TWO_STREAMS:
            g_timed, timer.rec = timer.rec, []
            _mc._TWO_STREAMS = False
            step.step(batch)
            timer.on = True
            for _ in range(2):
                step.step(batch)
            torch.cuda.synchronize()
            timer.on = False
            _mc._TWO_STREAMS = True
            g_excl = timer.summary()
            timer.rec = g_timed

    if rank == 0:
        pairs = world * cfg["batch"] * args.steps
        ms = 1e3 * elapsed / args.steps
        fpp = algorithmic_flops_per_pair(cfg)
        g = timer.summary()
        out = {
            "metric": "image-text pairs/sec/step (two-tower+ArcFace, bs=256)", "value": pairs / elapsed, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: " + {"multimodal": f"roberta-wwm-ext-{cfg.get('text')} + {cfg.get('image')} two-tower + ArcFace",
                                                        "nlp": "text tower + ArcFace", "cv": "image tower + ArcFace"}[cfg["kind"]],
                       "per_gpu_batch": cfg["batch"], "global_batch": world * cfg["batch"], "seq_len": cfg.get("seq_len"),
                       # what the process group actually is: ranks RCCL (backend "nccl" on ROCm) connected, one per GPU
                       "dist_backend": dist.get_backend() if world > 1 else None,
                       "ranks_in_process_group": dist.get_world_size() if world > 1 else 1,
                       "head": type(model.classifier).__name__,
                       "grad_exchange_dtype": (str(step.exchange.grad_dtype).replace("torch.", "") if step.exchange else None),
                       "image": cfg.get("res"), "classes": cfg["classes"], "parallelism": f"dp{world}",
                       "dropout": not args.no_dropout, "attention_mask": "ragged U{8..S}" if args.ragged_masks else "all ones", "loss_path": "literal" if args.literal_loss else "fused",
                       "algorithmic_gflop_per_pair": fpp / 1e9,
                       "step_mfma_frac": (fpp * cfg["batch"] / (ms * 1e-3) / 1e12) / MFMA_BF16_PEAK_TFLOPS,
                       "final_loss": lossv, "launch": launch,
                       # the same step launched kernel by kernel from Python (what N > 1 ranks run), same process / box, after the timed region
                       "eager_ms_per_step": eager_ms,
                       # every MMSIM_* switch set in this process (they select schedules / kernels): empty = the defaults
                       "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("MMSIM_")}},
            "roofline": None if g is None else {
                "bound": "mfma", "kernel": "gemm_pp64_kernel<false,true,256> (Y = X W^T: 256x256 tiles, 64-deep LDS-DMA slices, ping-pong wave groups, bf16 MFMA 16x16x32, fp32 accumulate)",
                "achieved": g["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": g["tflops"] / MFMA_BF16_PEAK_TFLOPS,
                "traffic": _pmc_traffic(args.config)[0], "traffic_unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/r04_pmc_gemm_pp64.json, re-measured with this round's kernel)",
                "algorithmic_bytes_per_launch": _pmc_traffic(args.config)[1],
                "launches": g["launches"], "avg_launch_us": g["avg_us"], "measured_over": roof_from,
                "concurrent_with": "image-tower kernels on a second stream" if g_excl else None,
                "achieved_exclusive": g_excl["tflops"] if g_excl else None,
                "frac_exclusive": g_excl["tflops"] / MFMA_BF16_PEAK_TFLOPS if g_excl else None},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_batch, args.cpu_steps)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
