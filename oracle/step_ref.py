"""Oracle: one full training step of the two-tower + ArcFace model on the CPU, fp32.  Test infrastructure only.

Composes the restatements (bert_ref, effnet_ref, arcface_ref) in the reference's step order
(multimodal_classifier_train.py:177-201) with the exact optimiser objects the reference builds
(torch.optim.AdamW; linear schedules per optim_ref.linear_lr, :152-164).  Used by tests (parity of the loss
curve) and by bench.py's ``cpu_baseline`` leg (kind "port"), never by the product path.
"""
import torch

from . import arcface_ref, bert_ref, effnet_ref, optim_ref


class TwoTowerOracle:
    def __init__(self, text_shape=None, text_state=None, image_name=None, image_state=None, head_weight=None,
                 num_steps=1000, use_fc=False, margin=0.5, lr_emb=5e-5, lr_fc=1e-2, warmup_fc=0.15):
        self.text_shape, self.image_name, self.use_fc, self.margin = text_shape, image_name, use_fc, margin
        leaf = lambda t: t.detach().clone().float().requires_grad_(True)
        self.text = {k: leaf(v) for k, v in text_state.items()} if text_state is not None else None
        self.image = ({k: (leaf(v) if v.is_floating_point() and "running" not in k else v.clone())
                       for k, v in image_state.items()} if image_state is not None else None)
        self.head = leaf(head_weight)
        tower = ([v for v in self.text.values()] if self.text else []) + \
                ([v for v in self.image.values() if torch.is_tensor(v) and v.requires_grad] if self.image else [])
        self.opt_emb = torch.optim.AdamW(tower, lr=lr_emb)
        self.opt_fc = torch.optim.AdamW([self.head], lr=lr_fc)
        self.lr_emb0, self.lr_fc0, self.total, self.warm = lr_emb, lr_fc, num_steps, warmup_fc * num_steps
        self.t = 0
        self._set_lr()

    def _set_lr(self):
        self.opt_emb.param_groups[0]["lr"] = optim_ref.linear_lr(self.lr_emb0, self.t, 0, self.total)
        self.opt_fc.param_groups[0]["lr"] = optim_ref.linear_lr(self.lr_fc0, self.t, self.warm, self.total)

    def embed(self, batch):
        embs = []
        if self.image is not None:
            embs.append(effnet_ref.cv_predict_emb(self.image, self.image_name, batch["img_tensor"], use_fc=self.use_fc, training=True))
        if self.text is not None:
            embs.append(bert_ref.bert_forward(self.text, self.text_shape, batch["input_ids"], batch.get("token_type_ids"),
                                              batch.get("attention_mask")))
        return arcface_ref.glue_concat(*embs) if len(embs) == 2 else embs[0]

    def step(self, batch):
        emb = self.embed(batch)
        logits = arcface_ref.arcface_forward(emb, self.head, batch["labels"], 64.0, self.margin)
        loss = arcface_ref.ce_loss(logits, batch["labels"])
        loss.backward()
        pred = logits.argmax(1)
        # multimodal_classifier_train.py:195-201: optimizer_emb.step(); lr_scheduler_emb.step(); optimizer_fc.step();
        # lr_scheduler_fc.step() -- both updates of step t use lr(t)
        self.opt_emb.step()
        self.opt_emb.zero_grad()
        self.opt_fc.step()
        self.opt_fc.zero_grad()
        self.t += 1
        self._set_lr()
        return loss.detach(), pred
