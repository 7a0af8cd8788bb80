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


SHARD_HEAD_FROM = 500000      # classes from which a data-parallel run shards the head by default (BASELINE config 5: 1 M)


def _want_sharded_head(cfg):
    """``cfg['sharded_head']`` forces it on / off; by default a data-parallel run shards the head from SHARD_HEAD_FROM classes."""
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    want = cfg.get("sharded_head")
    if want is None:
        want = cfg["classes"] >= SHARD_HEAD_FROM
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
    c = SimpleNamespace(**{k: v for k, v in cfg.items() if k != "sharded_head"})
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

    def _set_lr(self):
        self.opt_emb.param_groups[0]["lr"] = linear_schedule_lr(self.lr_emb0, self.t, 0, self.total)
        self.opt_fc.param_groups[0]["lr"] = linear_schedule_lr(self.lr_fc0, self.t, self.warmup_fc, self.total)

    def step(self, batch):
        """One training step; returns (loss tensor, argmax predictions) still on the device (no host sync)."""
        model = self.model
        model.train()
        kw = model_inputs(self.kind, batch)
        if self.fused_loss:
            loss, pred = model.forward_loss(**kw)
        else:                                   # the reference's literal path: materialised logits + nn.CrossEntropyLoss
            logits = model(**kw)
            loss = self.ce(logits, batch["labels"])
            pred = torch.argmax(logits, dim=-1)
        loss.backward()
        if self.exchange:
            self.exchange.finish()
        # reference order (multimodal_classifier_train.py:195-201): BOTH optimisers step with lr(t); each schedule
        # advances right after its own optimiser, so the head's first warm-up step runs at lr 0 and the last step at lr(T-1)
        self.opt_emb.step()
        self.opt_emb.zero_grad()
        self.opt_fc.step()
        self.opt_fc.zero_grad()
        self.t += 1
        self._set_lr()                          # lr_scheduler_emb.step(); lr_scheduler_fc.step()
        return loss.detach(), pred


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
