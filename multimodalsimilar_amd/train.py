"""Training-step assembly shared by the entry points (multimodal_classifier_train.py, nlp_classifier_train.py),
bench.py and the tests: model construction from local configs, synthetic batches, and the step itself in the
reference's order (multimodal_classifier_train.py:177-201):

    model.train(); preds/loss = model(...); loss.backward();
    optimizer_emb.step(); lr_scheduler_emb.step(); optimizer_emb.zero_grad();
    optimizer_fc.step();  lr_scheduler_fc.step();  optimizer_fc.zero_grad()

Two optimisers as in the reference: AdamW(lr 5e-5) over both towers with a linear schedule without warm-up, and
AdamW(lr 1e-2) over the ArcFace head with 15 % (float) warm-up (:152-164) -- here as fused HIP launches over the
flat parameter buffers.  Under data parallelism the flat gradient buffers are all-reduced (sum) over RCCL while
backward is still running and the 1/world factor is folded into the AdamW kernels.
"""
import os
import warnings
from types import SimpleNamespace

import torch
import torch.distributed as dist

from .bert import BertConfig, BertModel
from .dist import GradientExchange
from .optim import FusedAdam, FusedAdamW, linear_schedule_lr

VOCAB = 21128          # hfl/chinese-roberta-wwm-ext(-large)

CONFIGS = {
    # BASELINE.json configs[0..4]
    "cfg1": dict(kind="nlp", text="base", seq_len=64, batch=8, classes=1000),
    "cfg2": dict(kind="cv", image="efficientnet_b0", res=224, batch=128, classes=10000, fc_dim=512, use_fc=True),
    "cfg3": dict(kind="nlp", text="large", seq_len=128, batch=256, classes=10000),
    "cfg4": dict(kind="multimodal", text="large", image="efficientnet_b4", res=224, seq_len=128, batch=256, classes=100000,
                 use_fc=False),
    "cfg5": dict(kind="multimodal", text="large", image="efficientnet_b4", res=224, seq_len=128, batch=256, classes=1000000,
                 use_fc=False),
    # small shapes for smoke tests
    "tiny": dict(kind="multimodal", text="tiny", image="efficientnet_b0", res=64, seq_len=32, batch=8, classes=96,
                 use_fc=False),
}


def text_config(name, dropout=True):
    p = {} if dropout else dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    if name == "base":
        return BertConfig(vocab_size=VOCAB, **p)
    if name == "large":
        return BertConfig.roberta_wwm_ext_large(vocab_size=VOCAB, **p)
    if name == "tiny":
        return BertConfig(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                          max_position_embeddings=64, **p)
    raise ValueError(name)


# classes from which a data-parallel run shards the head by default (BASELINE config 5: 1 M).  MMSIM_SHARD_HEAD_FROM moves the
# threshold (bench.py --shard-head-from): at cfg4 the replicated head's 1.1 GB gradient is 45 % of the bytes all-reduced per step,
# with --shard-head-from 100000 it is exchanged as [B N, D] embeddings + [B N] row statistics instead.
SHARD_HEAD_FROM = int(os.environ.get("MMSIM_SHARD_HEAD_FROM", "500000"))


def _want_sharded_head(cfg):
    """``cfg['sharded_head']`` forces it on / off; by default a data-parallel run shards the head from SHARD_HEAD_FROM classes."""
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    want = cfg.get("sharded_head")
    if want is None:
        want = cfg["classes"] >= cfg.get("shard_head_from", SHARD_HEAD_FROM)
    return bool(want and on)


def build_model(cfg, device, seed=0, dropout=True):
    """Random-init model of a BASELINE config from local config objects (there are no checkpoints offline).
    Data parallelism with very many classes: the replicated ArcFace head is replaced by the class-sharded one (each rank owns
    C / world class rows; SURVEY.md H2-B, multimodalsimilar_amd/sharded_head.py)."""
    if not _want_sharded_head(cfg):
        return _build_model(cfg, device, seed, dropout)
    from .sharded_head import ShardedArcMarginProduct
    C = cfg["classes"]
    small = C * 4096 <= (1 << 28)       # small enough to draw the full matrix: the shards then equal the replicated init
    model = _build_model(cfg if small else dict(cfg, classes=8), device, seed, dropout)      # else: a stub head, never used
    old = model.classifier
    model.classifier = ShardedArcMarginProduct(old.in_feature, C, s=old.s, m=old.m, easy_margin=old.easy_margin, seed=seed,
                                               full_weight=old.weight.detach().cpu() if small else None).to(device)
    model.num_labels = C
    return model


def _build_model(cfg, device, seed=0, dropout=True):
    from nlp_classifier import NlpClassifier
    from cv_classifier import CvClassifier
    from multimodal_classifier import MultimodalClassifier
    torch.manual_seed(seed)
    c = SimpleNamespace(**{k: v for k, v in cfg.items() if k not in ("sharded_head", "shard_head_from")})
    if c.kind == "nlp":
        return NlpClassifier(BertModel(text_config(c.text, dropout), seed=seed), num_labels=c.classes).to(device)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cv = CvClassifier(c.image, getattr(c, "fc_dim", 512), c.classes, pretrained=False, use_fc=c.use_fc)
    if c.kind == "cv":
        return cv.to(device)
    nlp = NlpClassifier(BertModel(text_config(c.text, dropout), seed=seed), num_labels=c.classes)
    emb = (cv.fc.out_features if c.use_fc else cv.backbone.num_features) + nlp.ptm.config.hidden_size
    return MultimodalClassifier(device, cv, nlp, emb_size=emb, num_labels=c.classes)


def synthetic_batch(cfg, device, seed=1234, batch=None, vocab=None, ragged_masks=False):
    """images ~ N(0,1) fp32 NCHW, ids uniform with [CLS]=101 at position 0, token types 0,
    all-ones attention mask (the roofline run) or, with ragged_masks, per-row lengths ~ U{8..S} zero-padded to S as the
    reference pads to max_length (multimodal_dataset.py:44-48; the realism run), labels uniform."""
    c = SimpleNamespace(**cfg)
    B = batch or c.batch
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    if c.kind in ("cv", "multimodal"):
        out["img_tensor"] = torch.randn(B, 3, c.res, c.res, generator=g).to(device)
    if c.kind in ("nlp", "multimodal"):
        V = vocab or (512 if c.text == "tiny" else VOCAB)
        ids = torch.randint(0, V, (B, c.seq_len), generator=g)
        ids[:, 0] = 101
        out["input_ids"] = ids.to(device)
        out["token_type_ids"] = torch.zeros(B, c.seq_len, dtype=torch.long, device=device)
        if ragged_masks:
            lens = torch.randint(min(8, c.seq_len), c.seq_len + 1, (B,), generator=g)
            out["attention_mask"] = (torch.arange(c.seq_len).unsqueeze(0) < lens.unsqueeze(1)).long().to(device)
        else:
            out["attention_mask"] = torch.ones(B, c.seq_len, dtype=torch.long, device=device)
    out["labels"] = torch.randint(0, c.classes, (B,), generator=g).to(device)
    return out


def model_inputs(kind, batch):
    if kind == "nlp":
        return dict(query_input_ids=batch["input_ids"], query_token_type_ids=batch["token_type_ids"],
                    query_attention_mask=batch["attention_mask"], label=batch["labels"])
    if kind == "cv":
        return dict(input=batch["img_tensor"], label=batch["labels"])
    return dict(img_input=batch["img_tensor"], query_input_ids=batch["input_ids"],
                query_token_type_ids=batch["token_type_ids"], query_attention_mask=batch["attention_mask"],
                label=batch["labels"])


class TrainStep:
    """Optimisers + schedules + (optional) data-parallel gradient exchange around a model, stepped in the reference order."""

    def __init__(self, model, kind, num_training_steps, lr_emb=5e-5, lr_fc=1e-2, warmup_fc=0.15, fused_loss=True):
        self.model, self.kind, self.fused_loss = model, kind, fused_loss
        self.total = num_training_steps
        if kind == "multimodal":
            towers = [model.cv, model.nlp]                  # multimodal_classifier_train.py:152-156
        elif kind == "nlp":
            towers = [model.emb_layer]                      # nlp_classifier_train.py:89
        else:
            towers = [model]                                # image-only: everything except the head
        self.exchange = GradientExchange(model) if dist.is_initialized() and dist.get_world_size() > 1 else None
        gs = self.exchange.grad_scale if self.exchange else 1.0
        # the towers' private ArcFace heads (cv.classifier / nlp.classifier) never receive a gradient on the two-tower
        # path; FusedAdamW skips gradient-less buffers exactly as torch skips parameters whose .grad is None
        self.opt_emb = FusedAdamW(towers, lr=lr_emb, grad_scale=gs, exclude=[model.classifier])
        self.opt_fc = FusedAdamW(model.classifier, lr=lr_fc, grad_scale=gs)
        self.lr_emb0, self.lr_fc0, self.warmup_fc = lr_emb, lr_fc, warmup_fc * num_training_steps
        self.t = 0
        self._set_lr()
        self.ce = torch.nn.CrossEntropyLoss()
        self._init_opt_in_backward()

    # ---- the optimiser updates issued DURING the backward (single process; MMSIM_OPT_IN_BWD=1 turns it on, default OFF).  AdamW is a
    # pure HBM stream (3.4 ms per step at cfg4 with the matrix pipes idle) and the text tower's backward a chain of MFMA-bound products
    # that leave HBM idle: the head's update (its gradient is final before the towers' backward starts) and each encoder layer's
    # range run on their own stream beside those products.  Same launches on the same data -- bit-identical to stepping at the end
    # (tests/test_gpu_graph_step.py).  MEASURED (round 4, cfg4, alternating runs on one box).  Replayed hipGraph: 98.2 / 98.2 ms per
    # step against 92.3 / 92.1 with the updates at the end -- 6 ms SLOWER; with the towers' ranges only ("2", no head update) 98.9
    # against 93.2, and capping the update's grid at 256 / 1 024 blocks (one wave per SIMD, which fits beside a GEMM workgroup) changes
    # nothing (98.4 / 98.7).  EAGER launches of the same step: 93.3 against 93.7 -- a small gain, the size the co-residency
    # experiment predicts (a third of the 1.9 ms moved).  So the loss is the replayed graph's: with ~26 more fork / join edges per step
    # its branches are no longer scheduled as two long streams.  Off; kept for eager (data-parallel) runs to try.
    def _init_opt_in_backward(self):
        self._oib = None
        dev_ok = any(f.master.is_cuda for f in self.opt_emb.flats)
        mode = os.environ.get("MMSIM_OPT_IN_BWD", "0")
        if self.exchange is not None or not dev_ok or mode not in ("1", "2"):
            return
        from .bert import BertModel
        from .head import ArcMarginProduct
        owner = {id(f): self.opt_emb for f in self.opt_emb.flats}
        owner.update({id(f): self.opt_fc for f in self.opt_fc.flats})
        kinds = (BertModel, ArcMarginProduct) if mode == "1" else (BertModel,)        # "2": the towers' ranges only
        mods = [m for m in self.model.modules() if isinstance(m, kinds) and hasattr(m, "grad_ready_hook")
                and all(id(f) in owner for f in m.flat_buffers())]
        if not mods:
            return
        self._oib = dict(owner=owner, stream=torch.cuda.Stream(), active=False)
        for m in mods:
            m.grad_ready_hook = self._on_grad_ready

    def _on_grad_ready(self, flat, lo, hi):
        o = self._oib
        if not o["active"]:
            return
        ev = torch.cuda.Event()
        ev.record()                               # behind every kernel of the reporting tower that reads or writes this range
        o["stream"].wait_event(ev)
        with torch.cuda.stream(o["stream"]):
            o["owner"][id(flat)].step_range(flat, lo, hi)

    def _set_lr(self):
        self.opt_emb.param_groups[0]["lr"] = linear_schedule_lr(self.lr_emb0, self.t, 0, self.total)
        self.opt_fc.param_groups[0]["lr"] = linear_schedule_lr(self.lr_fc0, self.t, self.warmup_fc, self.total)

    def _body(self, batch):
        """The device work of one step: forward, loss, backward, gradient exchange, both optimiser updates (reference order)."""
        model = self.model
        model.train()
        kw = model_inputs(self.kind, batch)
        if self.fused_loss:
            loss, pred = model.forward_loss(**kw)
        else:                                   # the reference's literal path: materialised logits + nn.CrossEntropyLoss
            logits = model(**kw)
            loss = self.ce(logits, batch["labels"])
            pred = torch.argmax(logits, dim=-1)
        if self._oib:
            self.opt_emb.begin_ranged_step(); self.opt_fc.begin_ranged_step()
            self._oib["active"] = True
        try:
            loss.backward()
        finally:
            if self._oib:
                self._oib["active"] = False
        if self.exchange:
            self.exchange.finish()
        # reference order (multimodal_classifier_train.py:195-201): BOTH optimisers step with lr(t); each schedule
        # advances right after its own optimiser, so the head's first warm-up step runs at lr 0 and the last step at lr(T-1)
        if self._oib:
            torch.cuda.current_stream().wait_stream(self._oib["stream"])     # the ranged updates are part of this step
            self.opt_emb.finish_ranged_step()
            self.opt_emb.zero_grad()
            self.opt_fc.finish_ranged_step()
        else:
            self.opt_emb.step()
            self.opt_emb.zero_grad()
            self.opt_fc.step()
        self.opt_fc.zero_grad(lazy=True)        # the head's dW product overwrites its whole buffer next step (flat.zero_grad)
        return loss.detach(), pred

    def _advance(self):
        self.t += 1
        self._set_lr()                          # lr_scheduler_emb.step(); lr_scheduler_fc.step()

    def step(self, batch):
        """One training step; returns (loss tensor, argmax predictions) still on the device (no host sync)."""
        out = self._body(batch)
        self._advance()
        return out


class GraphedTrainStep:
    """The training step as ONE hipGraph: captured once, replayed every step.

    Why: a step is ~2 500 kernel launches issued from Python at ~40 us each -- ~105 ms of host time, as long as the step itself.
    The image tower's backward (~700 short launches) could not even START until the host had finished enqueueing the text
    tower's (~30 ms), and then ran at the host's launch rate, not the GPU's (rocprofv3 timeline: the image stream idle for 30 ms,
    then 15 ms of exposed tail).  Replaying a captured graph costs the host ~nothing, so the two towers' streams really overlap.

    What a captured graph freezes, and how each is kept live:
      * kernel ARGUMENTS that change per step -- the AdamW learning rates / bias corrections and the dropout seeds: the kernels
        read them from device memory instead (FusedAdamW.dev_hyper, mmsim_set_step_seed_ptr), which this class refreshes with
        two small host-to-device copies before every replay;
      * tensor addresses: activations come from the graph's private memory pool (stable across replays), the batch is copied
        into static input tensors;
      * host-side decisions (which kernel, which split-K, the margin): fixed per shape -- ``update_m`` or a new batch shape need a
        new capture (``recapture()``).
    Single-process only (collectives are left to the eager step).  Semantics are TrainStep.step's, checked against it step by
    step in tests/test_gpu_graph_step.py."""

    def __init__(self, ts, batch, warmup=2):
        from .optim import hyper_values
        from ._lib import lib
        if ts.exchange is not None:
            raise ValueError("GraphedTrainStep: data-parallel runs use the eager TrainStep")
        self.ts, self._hv, self._lib = ts, hyper_values, lib
        dev = batch["labels"].device
        self.static = {k: v.clone() for k, v in batch.items()}
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)         # [0:3] towers, [4:7] head (16-byte aligned halves)
        self.seed = torch.zeros(1, dtype=torch.int64, device=dev)
        self._init_host_ring()
        self.replays = 0
        ts.opt_emb.dev_hyper, ts.opt_fc.dev_hyper = self.hyper[0:3], self.hyper[4:7]
        lib.set_step_seed_ptr(self.seed.data_ptr())
        self.graph = None
        self._capture(warmup)

    RING = 64

    def _init_host_ring(self):
        # The host runs AHEAD of the device (that is the point of replaying a graph): a single pinned staging buffer would be
        # overwritten with step k+1's values before step k's asynchronous copy has read it.  A ring of pinned rows, each guarded
        # by the event recorded behind the copies that read it.
        self._h_hyper = torch.zeros(self.RING, 8, dtype=torch.float32).pin_memory()
        self._h_seed = torch.zeros(self.RING, 1, dtype=torch.int64).pin_memory()
        self._h_done = [None] * self.RING

    def _prepare(self):
        """Write the NEXT step's scalars (lr and bias corrections of both optimisers, the dropout step word) to the device."""
        ts = self.ts
        slot = self.replays % self.RING
        if self._h_done[slot] is not None:
            self._h_done[slot].synchronize()             # the copies that read this row 64 steps ago have run
        self._h_hyper[slot, 0:3] = torch.tensor(self._hv(ts.opt_emb, ts.opt_emb._t + 1))
        self._h_hyper[slot, 4:7] = torch.tensor(self._hv(ts.opt_fc, ts.opt_fc._t + 1))
        self._h_seed[slot, 0] = (self.replays * 0x9E3779B97F4A7C15 + 0x5851F42D4C957F2D) & 0x7FFFFFFFFFFFFFFF
        self.hyper.copy_(self._h_hyper[slot], non_blocking=True)
        self.seed.copy_(self._h_seed[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._h_done[slot] = ev
        self.replays += 1

    def _capture(self, warmup):
        ts = self.ts
        side = torch.cuda.Stream()                       # torch: warm up on a side stream before capturing
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                      # real (eager) steps: lazy initialisation, scratch buffers, autotuned choices
                self._prepare()
                ts._body(self.static)
                ts._advance()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        t_emb, t_fc = ts.opt_emb._t, ts.opt_fc._t
        # the gradient state every replay starts from: zero, and -- for a buffer whose only writer overwrites (flat.zero_grad) --
        # marked so; without this a capture taken before any eager step would record the head's dW product as an accumulation
        for f in list(ts.opt_emb.flats) + list(ts.opt_fc.flats):
            f.materialize_zero()
            f.zero_grad(lazy=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.pred = ts._body(self.static)
        ts.opt_emb._t, ts.opt_fc._t = t_emb, t_fc        # capture records, it does not run: undo the host-side step counters

    def recapture(self, warmup=0):
        self.graph = None
        self._capture(warmup)

    def step(self, batch):
        for k, v in batch.items():
            if v is not self.static[k]:
                self.static[k].copy_(v, non_blocking=True)
        self._prepare()
        self.graph.replay()
        ts = self.ts
        ts.opt_emb._t += 1
        ts.opt_fc._t += 1
        ts._advance()
        return self.loss, self.pred

    def close(self):
        self._lib.set_step_seed_ptr(None)
        self.ts.opt_emb.dev_hyper = self.ts.opt_fc.dev_hyper = None


class CvTrainLoop:
    """The reference's image-only training loop around a ``CvClassifier`` (cv_classifier_train_daodian.py):

        optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)                                   (:264)
        scheduler = CosineAnnealingWarmRestarts(optimizer, T_0=7, T_mult=1, eta_min=1e-6)           (:267)
        per step :  optimizer.zero_grad(); output = model(images, targets); loss = CE(output, targets);
                    loss.backward(); optimizer.step()                                               (:108-142)
        per epoch:  scheduler.step()  (:139);  model.classifier.update_m(0.04)                      (:292)

    ONE optimiser over every parameter (backbone, fc / bn top and the ArcFace head), as ``model.parameters()`` is there.  The
    scheduler is torch's own object driving the fused optimiser's param group; the margin is a kernel argument, so annealing it
    costs nothing (no rebuild, no recompilation).  Under data parallelism gradients are exchanged as in TrainStep."""

    def __init__(self, model, lr=1e-3, T_0=7, T_mult=1, eta_min=1e-6, margin_step=0.04, fused_loss=True):
        self.model, self.fused_loss, self.margin_step = model, fused_loss, margin_step
        self.exchange = GradientExchange(model) if dist.is_initialized() and dist.get_world_size() > 1 else None
        gs = self.exchange.grad_scale if self.exchange else 1.0
        self.optimizer = FusedAdam(model, lr=lr, grad_scale=gs)
        self.scheduler = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(self.optimizer, T_0=T_0, T_mult=T_mult, eta_min=eta_min)
        self.ce = torch.nn.CrossEntropyLoss()
        self.epoch = 0

    def step(self, batch):
        """One training step; returns (loss, argmax of the margin logits) on the device."""
        model = self.model
        model.train()
        self.optimizer.zero_grad()
        if self.fused_loss:
            loss, pred = model.forward_loss(batch["img_tensor"], batch["labels"])
        else:
            output = model(batch["img_tensor"], batch["labels"])
            loss = self.ce(output, batch["labels"])
            pred = torch.argmax(output, 1)
        loss.backward()
        if self.exchange:
            self.exchange.finish()
        self.optimizer.step()
        return loss.detach(), pred

    def end_epoch(self):
        self.scheduler.step()                              # :139
        self.model.classifier.update_m(self.margin_step)   # :292
        self.epoch += 1

    def evaluate(self, batch):
        """Validation step (:146-170): cosine logits (``is_test=True``), their argmax and the cross-entropy the loop logs."""
        self.model.eval()
        with torch.no_grad():
            output = self.model(batch["img_tensor"], batch["labels"], is_test=True)
            return self.ce(output, batch["labels"]), torch.argmax(output, 1)
