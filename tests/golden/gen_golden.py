"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE in the build container.

Run once, here (CPU):   python tests/golden/gen_golden.py
It imports /root/reference's ``arcface`` and ``nlp_classifier`` (over HF ``BertModel`` built from a
local config object - no downloads) and the exact optimiser/scheduler objects the reference's
train script constructs, feeds them seeded synthetic inputs and stores inputs + outputs as .npz.
Only the numeric vectors are committed; no reference source travels.
cv_classifier / multimodal_classifier cannot be imported as they are (ModuleNotFoundError: timm; torchvision is absent too).
glue_0.npz composes the reference's ArcMarginProduct(m=0.5) with the three glue operations of multimodal_classifier.py:54-56
applied to stand-in tower embeddings; multimodal_forward.npz (round 3) EXECUTES multimodal_classifier.py
itself -- its __init__ (torch.load of two pickled towers), forward, forward(is_test=True) and predict_emb, lines 14-57 -- with
empty stub modules named timm / torchvision placed in sys.modules INSIDE THIS GENERATOR ONLY (the reference imports them at
module level and never uses them in that file), a small stand-in image tower exposing predict_emb, and the reference's own
NlpClassifier over a tiny HF BertModel as the text tower.
"""
import os
import sys
import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
os.chdir(REF)
torch.set_num_threads(4)

import arcface as ref_arcface            # noqa: E402
import nlp_classifier as ref_nlp         # noqa: E402
from transformers import BertConfig, BertModel, get_scheduler  # noqa: E402


def npz(name, **kw):
    np.savez(os.path.join(OUT, name), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                         for k, v in kw.items()})
    print("wrote", name)


def gen_arcface():
    cases = [(4, 32, 10, 0.5, False), (8, 64, 50, 0.4, False), (8, 64, 50, 0.2, True), (16, 128, 1000, 0.5, False),
             (32, 256, 777, 0.5, False)]
    for i, (B, D, C, m, easy) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + i)
        head = ref_arcface.ArcMarginProduct(in_feature=D, out_feature=C, s=64.0, m=m, easy_margin=easy)
        with torch.no_grad():
            head.weight.copy_(torch.randn(C, D, generator=g) * 0.1)
        x = torch.randn(B, D, generator=g)
        label = torch.randint(0, C, (B,), generator=g)
        with torch.no_grad():
            # force both margin branches: row 0 nearly opposite to its class row (cos <= th -> cos - mm),
            # row 1 nearly aligned with it (cos > th -> cos(theta+m)), row 2 mildly negative (easy-margin else)
            x[0] = -head.weight[label[0]] * 3 + 0.05 * torch.randn(D, generator=g)
            x[1] = head.weight[label[1]] * 2 + 0.05 * torch.randn(D, generator=g)
            x[2] = -0.3 * head.weight[label[2]] + 0.3 * torch.randn(D, generator=g)
        x.requires_grad_(True)
        logits = head(x, label)
        loss = torch.nn.CrossEntropyLoss()(logits, label)
        loss.backward()
        logits_test = head.forward_test(x.detach())
        npz(f"arcface_{i}.npz", x=x, weight=head.weight, label=label, s=64.0, m=m, easy=int(easy),
            logits=logits, logits_test=logits_test, loss=loss, dx=x.grad, dw=head.weight.grad)
    # update_m trajectory (arcface.py:35-42)
    head = ref_arcface.ArcMarginProduct(8, 4, m=0.2)
    traj = []
    for d in [0.04] * 25 + [-2.0, 0.5]:
        head.update_m(d)
        traj.append([head.m, head.cos_m, head.sin_m, head.th, head.mm])
    npz("arcface_update_m.npz", deltas=np.array([0.04] * 25 + [-2.0, 0.5]), traj=np.array(traj))


def gen_nlp(only_cfg=None):
    cfgs = {
        "tiny": dict(vocab_size=128, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=512, max_position_embeddings=64, B=4, S=32, C=40, store_weights=True),
        "mid": dict(vocab_size=256, hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                    intermediate_size=1024, max_position_embeddings=128, B=2, S=128, C=300, store_weights=False),
        # BASELINE config 1's shape (roberta-base: H 768, 12 heads, FFN 3072, vocab 21128, S 64, B 8, 1000 classes), ONE layer
        # (the state is re-seeded on the test side, only vectors and gradient norms are stored)
        "base1": dict(vocab_size=21128, hidden_size=768, num_hidden_layers=1, num_attention_heads=12,
                      intermediate_size=3072, max_position_embeddings=512, B=8, S=64, C=1000, store_weights=False),
    }
    if only_cfg is not None:
        cfgs = {only_cfg: cfgs[only_cfg]}
    sys.path.insert(0, os.path.abspath(os.path.join(OUT, "..", "..")))
    from oracle import bert_ref
    for name, c in cfgs.items():
        torch.manual_seed(7)
        conf = BertConfig(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"],
                          num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                          intermediate_size=c["intermediate_size"], max_position_embeddings=c["max_position_embeddings"],
                          hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        conf._attn_implementation = "eager"
        ptm = BertModel(conf)
        shape = bert_ref.BertShape(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"],
                                   num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                                   intermediate_size=c["intermediate_size"],
                                   max_position_embeddings=c["max_position_embeddings"])
        sd0 = bert_ref.init_state(shape, seed=11)
        # non-trivial LN / bias values so every term of the backward is exercised
        g = torch.Generator().manual_seed(12)
        for k in sd0:
            if k.endswith("LayerNorm.weight"):
                sd0[k] = 1.0 + 0.1 * torch.randn(sd0[k].shape, generator=g)
            elif k.endswith(".bias"):
                sd0[k] = 0.05 * torch.randn(sd0[k].shape, generator=g)
        missing = ptm.load_state_dict(sd0, strict=False)
        assert not missing.missing_keys or all("position_ids" in k for k in missing.missing_keys), missing
        model = ref_nlp.NlpClassifier(ptm, num_labels=c["C"])
        model.train()   # dropout p=0 in config; NlpClassifier's own nn.Dropout is never applied (E6)
        with torch.no_grad():
            if name == "base1":      # re-seedable on the test side (the 3 MB head is not stored): its own generator
                model.classifier.weight.copy_(torch.randn(c["C"], c["hidden_size"], generator=torch.Generator().manual_seed(13)) * 0.05)
            else:
                model.classifier.weight.copy_(torch.randn(c["C"], c["hidden_size"], generator=g) * 0.05)
        B, S = c["B"], c["S"]
        ids = torch.randint(0, c["vocab_size"], (B, S), generator=g)
        tt = torch.randint(0, 2, (B, S), generator=g)
        lens = torch.randint(S // 4, S + 1, (B,), generator=g)
        lens[0] = S
        mask = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).long()
        label = torch.randint(0, c["C"], (B,), generator=g)
        emb = model.predict_emb(ids, tt, None, mask)
        logits = model(ids, tt, None, mask, label)
        loss = torch.nn.CrossEntropyLoss()(logits, label)
        loss.backward()
        logits_test = model(ids, tt, None, mask, label, is_test=True)
        out = dict(input_ids=ids, token_type_ids=tt, attention_mask=mask, label=label, pooled=emb, logits=logits,
                   logits_test=logits_test, loss=loss, head_weight=model.classifier.weight,
                   head_grad=model.classifier.weight.grad, seed_state=11, seed_perturb=12)
        grads = {n: p.grad for n, p in ptm.named_parameters() if p.grad is not None}
        if c["store_weights"]:
            for k, v in sd0.items():
                out["w::" + k] = v
            for k, v in grads.items():
                out["g::" + k] = v
        else:
            for k, v in grads.items():
                out["gnorm::" + k] = v.norm()
            out["g::pooler.dense.weight"] = grads["pooler.dense.weight"]
            out["g::encoder.layer.0.attention.self.query.bias"] = grads["encoder.layer.0.attention.self.query.bias"]
            if name == "base1":      # a few full gradients, the 768 x 768 ones as float16 (relative 5e-4: far below the bounds they serve)
                del out["g::pooler.dense.weight"], out["head_weight"]
                out["seed_head"] = 13
                out["head_grad"] = out["head_grad"].detach().numpy().astype(np.float16)
                for k in ("encoder.layer.0.attention.self.value.weight", "encoder.layer.0.attention.output.dense.weight"):
                    out["g::" + k] = grads[k].numpy().astype(np.float16)
                for k in ("encoder.layer.0.intermediate.dense.bias", "encoder.layer.0.output.LayerNorm.weight"):
                    out["g::" + k] = grads[k]
                out["g64::embeddings.position_embeddings.weight"] = grads["embeddings.position_embeddings.weight"][:64]   # rows >= S are zero
        npz(f"nlp_{name}.npz", **out)


def gen_nlp_base1():
    gen_nlp("base1")


def gen_glue():
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    B, Di, Dt, C = 8, 48, 32, 25
    img = (torch.randn(B, Di, generator=g) * 3).requires_grad_(True)
    txt = torch.tanh(torch.randn(B, Dt, generator=g)).requires_grad_(True)
    head = ref_arcface.ArcMarginProduct(in_feature=Di + Dt, out_feature=C, m=0.5)   # multimodal_classifier.py:22
    with torch.no_grad():
        head.weight.copy_(torch.randn(C, Di + Dt, generator=g) * 0.1)
    label = torch.randint(0, C, (B,), generator=g)
    final = torch.cat((F.normalize(img, p=2, dim=1), F.normalize(txt, p=2, dim=1)), 1)   # :54-56
    logits = head(final, label)
    loss = torch.nn.CrossEntropyLoss()(logits, label)
    loss.backward()
    npz("glue_0.npz", img=img, txt=txt, weight=head.weight, label=label, final=final, logits=logits, loss=loss,
        dimg=img.grad, dtxt=txt.grad, dw=head.weight.grad)


class StandInImageTower(torch.nn.Module):
    """Stand-in for the pickled CvClassifier of multimodal_classifier.py:16 (timm is absent): anything with predict_emb.
    Global average pool of the image -> Linear(3, 48) -> tanh x 3.  Its arithmetic is restated on the test side from the stored
    weights; only the reference's OWN lines (14-57) are what the fixture pins."""

    def __init__(self):
        super().__init__()
        self.fc = torch.nn.Linear(3, 48)

    def predict_emb(self, img):
        return 3.0 * torch.tanh(self.fc(img.mean((2, 3))))


def gen_multimodal_forward():
    """multimodal_classifier.py:14-57 executed as is (stub timm / torchvision modules only satisfy its imports)."""
    import tempfile
    import types
    for name in ("timm", "timm.data", "timm.data.transforms_factory", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["timm.data"].resolve_data_config = None                      # names the reference imports and never calls here
    sys.modules["timm.data.transforms_factory"].create_transform = None
    sys.modules["timm"].data = sys.modules["timm.data"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    import multimodal_classifier as ref_mm
    sys.path.insert(0, os.path.abspath(os.path.join(OUT, "..", "..")))
    from oracle import bert_ref
    c = dict(vocab_size=128, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
             max_position_embeddings=64)
    torch.manual_seed(7)
    conf = BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **c)
    conf._attn_implementation = "eager"
    ptm = BertModel(conf)
    shape = bert_ref.BertShape(**c)
    sd0 = bert_ref.init_state(shape, seed=11)                                # = the text tower of nlp_tiny.npz (same seeds)
    g = torch.Generator().manual_seed(12)
    for k in sd0:
        if k.endswith("LayerNorm.weight"):
            sd0[k] = 1.0 + 0.1 * torch.randn(sd0[k].shape, generator=g)
        elif k.endswith(".bias"):
            sd0[k] = 0.05 * torch.randn(sd0[k].shape, generator=g)
    ptm.load_state_dict(sd0, strict=False)
    nlp = ref_nlp.NlpClassifier(ptm, num_labels=5)                           # its own head is unused by the two-tower model
    g = torch.Generator().manual_seed(31)
    cv = StandInImageTower()
    with torch.no_grad():
        cv.fc.weight.copy_(torch.randn(48, 3, generator=g))
        cv.fc.bias.copy_(0.1 * torch.randn(48, generator=g))
    B, S, C, D = 6, 32, 30, 48 + 128
    with tempfile.TemporaryDirectory() as td:
        pc, pn = os.path.join(td, "cv.pt"), os.path.join(td, "nlp.pt")
        torch.save(cv, pc)
        torch.save(nlp, pn)
        real_load = torch.load
        torch.load = lambda f, *a, **k: real_load(f, *a, **{**k, "weights_only": False})     # whole-module pickles
        try:
            model = ref_mm.MultimodalClassifier("cpu", pc, pn, emb_size=D, num_labels=C)       # :14-25
        finally:
            torch.load = real_load
    model.train()
    assert abs(model.classifier.m - 0.5) < 1e-12 and model.classifier.s == 64.0
    with torch.no_grad():
        model.classifier.weight.copy_(torch.randn(C, D, generator=g) * 0.1)
    img = torch.randn(B, 3, 8, 8, generator=g)
    ids = torch.randint(0, c["vocab_size"], (B, S), generator=g)
    tt = torch.randint(0, 2, (B, S), generator=g)
    lens = torch.randint(S // 4, S + 1, (B,), generator=g)
    lens[0] = S
    mask = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).long()
    label = torch.randint(0, C, (B,), generator=g)
    final = model.predict_emb(img, ids, tt, None, mask)                                         # :44-57
    logits = model(img, ids, tt, None, mask, label)                                             # :27-40
    loss = torch.nn.CrossEntropyLoss()(logits, label)
    loss.backward()
    logits_test = model(img, ids, tt, None, mask, label, is_test=True)                          # :41-42
    grads = {n: p.grad for n, p in model.nlp.ptm.named_parameters() if p.grad is not None}
    # model.cv is the unpickled COPY of `cv` (multimodal_classifier.py:16): its gradients live there
    npz("multimodal_forward.npz", img=img, input_ids=ids, token_type_ids=tt, attention_mask=mask, label=label,
        cv_fc_weight=cv.fc.weight, cv_fc_bias=cv.fc.bias, head_weight=model.classifier.weight, final=final, logits=logits,
        logits_test=logits_test, loss=loss, head_grad=model.classifier.weight.grad, cv_fc_weight_grad=model.cv.fc.weight.grad,
        cv_fc_bias_grad=model.cv.fc.bias.grad, **{"g::pooler.dense.weight": grads["pooler.dense.weight"],
                                            "g::encoder.layer.1.output.dense.weight": grads["encoder.layer.1.output.dense.weight"]},
        seed_state=11, seed_perturb=12)


def gen_optim():
    g = torch.Generator().manual_seed(9)
    ps = [torch.nn.Parameter(torch.randn(5, 7, generator=g)), torch.nn.Parameter(torch.randn(11, generator=g)),
          torch.nn.Parameter(torch.randn(3, 2, 2, generator=g))]
    T = 10
    for tag, lr0, warm in (("emb", 5e-5, 0), ("fc", 1e-2, 0.15 * T)):
        params = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        opt = torch.optim.AdamW(params, lr=lr0)
        sch = get_scheduler(name="linear", optimizer=opt, num_warmup_steps=warm, num_training_steps=T)
        gg = torch.Generator().manual_seed(10)
        rec = dict(p0_0=params[0].detach().clone(), p0_1=params[1].detach().clone(), p0_2=params[2].detach().clone(),
                   lr0=lr0, warmup=warm, total=T)
        lrs = []
        for t in range(T):
            grads = [torch.randn(p.shape, generator=gg) for p in params]
            for p, gr in zip(params, grads):
                p.grad = gr.clone()
            lrs.append(opt.param_groups[0]["lr"])
            opt.step(); sch.step(); opt.zero_grad()
            for j, gr in enumerate(grads):
                rec[f"g{t}_{j}"] = gr
            for j, p in enumerate(params):
                rec[f"p{t + 1}_{j}"] = p.detach().clone()
        rec["lrs"] = np.array(lrs)
        npz(f"adamw_{tag}.npz", **rec)


def gen_multilabel():
    """nlp_classifier_multilabel.py: one embedding, three ArcFace heads (m = 0.4 / 0.2 / 0.1), and the weighted
    CE sum of nlp_classifier_train_daodian_v3_dist.py:164-166.  Text-tower weights = the ones of nlp_tiny.npz (same seeds)."""
    import nlp_classifier_multilabel as ref_ml
    sys.path.insert(0, os.path.abspath(os.path.join(OUT, "..", "..")))
    from oracle import bert_ref
    c = dict(vocab_size=128, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
             max_position_embeddings=64, B=4, S=32)
    torch.manual_seed(7)
    conf = BertConfig(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"],
                      num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
                      max_position_embeddings=c["max_position_embeddings"], hidden_dropout_prob=0.0,
                      attention_probs_dropout_prob=0.0)
    conf._attn_implementation = "eager"
    ptm = BertModel(conf)
    shape = bert_ref.BertShape(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"],
                               num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
                               max_position_embeddings=c["max_position_embeddings"])
    sd0 = bert_ref.init_state(shape, seed=11)
    g = torch.Generator().manual_seed(12)
    for k in sd0:
        if k.endswith("LayerNorm.weight"):
            sd0[k] = 1.0 + 0.1 * torch.randn(sd0[k].shape, generator=g)
        elif k.endswith(".bias"):
            sd0[k] = 0.05 * torch.randn(sd0[k].shape, generator=g)
    ptm.load_state_dict(sd0, strict=False)
    C = (7, 23, 61)
    model = ref_ml.NlpClassifierMultilabel(ptm, *C)
    model.train()
    g = torch.Generator().manual_seed(21)
    heads = (model.firstcate_classifier, model.secondcate_classifier, model.tag_classifier)
    with torch.no_grad():
        for h, n in zip(heads, C):
            h.weight.copy_(torch.randn(n, c["hidden_size"], generator=g) * 0.05)
    B, S = c["B"], c["S"]
    ids = torch.randint(0, c["vocab_size"], (B, S), generator=g)
    tt = torch.randint(0, 2, (B, S), generator=g)
    lens = torch.randint(S // 4, S + 1, (B,), generator=g)
    lens[0] = S
    mask = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).long()
    labels = [torch.randint(0, n, (B,), generator=g) for n in C]
    w = (1.0, 0.5, 2.0)
    logits = model(ids, tt, None, mask, labels[0], labels[1], labels[2])
    ce = torch.nn.CrossEntropyLoss()
    loss = sum(wi * ce(lg, y) for wi, lg, y in zip(w, logits, labels))
    loss.backward()
    test_logits = model(ids, tt, None, mask, is_test=True)
    out = dict(input_ids=ids, token_type_ids=tt, attention_mask=mask, weights=np.array(w), loss=loss,
               pooled=model.predict_emb(ids, tt, None, mask))
    for i, (h, y) in enumerate(zip(heads, labels)):
        out[f"label{i}"], out[f"head{i}"], out[f"head_grad{i}"] = y, h.weight, h.weight.grad
        out[f"logits{i}"], out[f"logits_test{i}"] = logits[i], test_logits[i]
    grads = {n: p.grad for n, p in ptm.named_parameters() if p.grad is not None}
    for k, v in grads.items():
        out["gnorm::" + k] = v.norm()
    out["g::pooler.dense.weight"] = grads["pooler.dense.weight"]
    out["g::encoder.layer.1.output.dense.weight"] = grads["encoder.layer.1.output.dense.weight"]
    npz("nlp_multilabel.npz", **out)


def gen_preprocess():
    """Input stage: Pillow itself (the library under torchvision's Resize in the reference's transform) resizes
    seeded uint8 images; the crop / ToTensor / Normalize steps of the published timm eval pipeline are applied with torch
    exactly as torchvision applies them.  Stored: inputs, Pillow's resized uint8 images, the final fp32 tensors."""
    import math
    from PIL import Image
    rng = np.random.default_rng(7)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    cases = [(37, 53, 32, 1.0), (64, 48, 32, 0.875), (75, 50, 64, 1.0), (40, 40, 64, 1.0), (131, 97, 32, 1.0)]   # H, W, S, crop_pct
    out = {"cases": np.array(cases, np.float64), "pillow": np.array([int(x) for x in Image.__version__.split(".")])}
    for i, (H, W, S, cp) in enumerate(cases):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        # smooth half of the cases: real photographs are not white noise, and clipping behaves differently on them
        if i % 2 == 0:
            img = np.asarray(Image.fromarray(img).resize((max(2, W // 8), max(2, H // 8))).resize((W, H), Image.BICUBIC))
        size = int(math.floor(S / cp))
        ow, oh = (size, int(size * H / W)) if W <= H else (int(size * W / H), size)          # torchvision Resize(int)
        r = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        top, left = int(round((oh - S) / 2.0)), int(round((ow - S) / 2.0))                    # torchvision CenterCrop
        c = torch.from_numpy(r[top:top + S, left:left + S].copy()).permute(2, 0, 1).contiguous()
        t = c.to(torch.float32).div(255)                                                       # ToTensor
        t = (t - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)           # Normalize
        out[f"img{i}"], out[f"resized{i}"], out[f"out{i}"] = img, r, t
    npz("preprocess.npz", **out)


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for fn in (gen_arcface, gen_nlp, gen_glue, gen_optim, gen_multilabel, gen_preprocess, gen_nlp_base1, gen_multimodal_forward):
        if only is None or fn.__name__ == "gen_" + only:
            fn()
