"""Inference-path parity: eval-mode image tower / two-tower embedding / cosine logits
against the CPU oracle on WELL-CONDITIONED weights -- BatchNorm running statistics equal to the batch statistics of the data
(what training leaves), small last-BatchNorm gamma in every residual block (what a trained residual network looks like) --
where running-statistic BatchNorm does not amplify rounding differences the way ~100 train-mode BatchNorms at random init do.
multimodal_classifier.py:44-57, cv_classifier.py:47-55, arcface.py:65-67.

The bound is north_star's: embeddings within 1e-2 (relative L2) for 16-bit storage.  Rounds 1-2 stored the image tower's
activations as bf16 and measured 4.5-5.9 % here (the fp32 oracle itself moves by 4.4-4.7 % when only its stored tensors are
rounded to bf16); since round 3 the forward tensors are fp16 (same bytes, 11-bit significand; csrc/common.h): the oracle under
fp16-storage emulation moves by d0 = 0.55-0.67 %, and the assertions below are the plain `e < 1e-2` again.  d0 is printed and
logged beside every measurement (tests/parity_log.py -> profiles/parity_r03.json)."""
import warnings

import pytest
import torch

from parity_log import check, record

pytestmark = pytest.mark.gpu
NORTH_STAR = 1e-2      # BASELINE.json north_star: outputs within 1e-2 for 16-bit storage
DEV = "cuda"


def l2err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def conditioned_cv(name, use_fc, seed, res, batch):
    """CvClassifier whose BatchNorms look trained: gamma / beta perturbed, the LAST BatchNorm of every residual block with a
    small gamma, running statistics := the fp32 batch statistics of a data batch (collected with the oracle)."""
    from oracle import effnet_ref
    from cv_classifier import CvClassifier
    warnings.simplefilter("ignore")
    torch.manual_seed(seed)
    model = CvClassifier(name, 64, 50, pretrained=False, use_fc=use_fc)
    g = torch.Generator().manual_seed(seed + 1)
    a = effnet_ref.arch(name)
    skip_last = {"backbone." + b["name"] + (".bn2" if b["type"] == "ds" else ".bn3") for b in a["blocks"] if b["skip"]}
    with torch.no_grad():
        for k, p in model.named_parameters():
            node = k.rsplit(".", 1)[0]
            if ".bn" in k or k.startswith("bn.") or k.startswith("backbone.bn"):
                if k.endswith("weight"):
                    p.copy_((0.25 if node in skip_last else 1.0) * (1.0 + 0.1 * torch.randn(p.shape, generator=g)))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
            if ".se." in k and p.dim() == 1:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
    if use_fc:
        model.dropout.p = 0.0
    x = torch.randn(batch, 3, res, res, generator=g)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    stats = {}
    effnet_ref.cv_predict_emb(sd, name, x, use_fc=use_fc, training=True, stats=stats)
    with torch.no_grad():
        for k, (mu, var) in stats.items():
            sd[k + ".running_mean"].copy_(mu)
            sd[k + ".running_var"].copy_(var)
    model.load_state_dict(sd)
    return model, sd, x


@pytest.mark.parametrize("name,use_fc,res", [("efficientnet_b0", True, 64), ("efficientnet_b4", False, 64), ("efficientnet_b4", True, 96)])
def test_image_tower_eval_mode_matches_the_oracle(name, use_fc, res):
    from oracle import effnet_ref, arcface_ref
    model, sd, x = conditioned_cv(name, use_fc, seed=3, res=res, batch=12)
    g = torch.Generator().manual_seed(9)
    xe = torch.randn(10, 3, res, res, generator=g)                     # fresh data through the running statistics
    ref = effnet_ref.cv_predict_emb(sd, name, xe, use_fc=use_fc, training=False)
    emu = effnet_ref.cv_predict_emb(sd, name, xe, use_fc=use_fc, training=False, emulate="fp16")
    model.to(DEV).eval()
    with torch.no_grad():
        emb = model.predict_emb(xe.to(DEV))
        cos = model(xe.to(DEV), is_test=True)
        emb2 = model.predict_emb(xe.to(DEV))
    e, d0 = l2err(emb, ref), l2err(emu, ref)
    cos_ref = arcface_ref.arcface_forward_test(ref, sd["classifier.weight"])
    ce = (cos.cpu() - cos_ref).abs().max().item()
    print(f"\n[{name} eval, fc={use_fc}, {res}^2] embedding L2 err {e:.4f} (oracle under fp16-storage emulation: {d0:.4f}); max |dcos| {ce:.4f}")
    assert torch.equal(emb, emb2)                                      # eval: deterministic
    tag = f"image_tower_eval_mode[{name}-fc{int(use_fc)}-{res}]"
    record(tag, "oracle under fp16-storage emulation: embedding relative L2 (d0)", d0, NORTH_STAR)
    check(tag, "embedding relative L2", e, NORTH_STAR)                  # north_star: 1e-2 for 16-bit storage
    check(tag, "max |cos - cos_ref| (cosine logits, unit scale)", ce, NORTH_STAR)
    # running statistics untouched by an eval forward
    assert torch.equal(model.backbone.bn1.running_mean.cpu(), sd["backbone.bn1.running_mean"])
    assert int(model.backbone.bn1.num_batches_tracked) == int(sd["backbone.bn1.num_batches_tracked"])


@pytest.mark.parametrize("name,use_fc,res", [("efficientnet_b0", True, 64), ("efficientnet_b4", False, 64)])
def test_image_tower_train_mode_on_conditioned_weights(name, use_fc, res):
    """Train-mode (batch-statistic BatchNorm) whole-tower forward + backward on the conditioned weights: embedding and loss within
    north_star's 1e-2; parameter gradients (bf16 gradient tensors through the whole tower) reported and bounded at 3 % median / 4 % p90 (measured 1.6-1.9 % / 2.1-2.3 %)."""
    from oracle import effnet_ref, arcface_ref
    model, sd, x = conditioned_cv(name, use_fc, seed=5, res=res, batch=16)
    y = torch.randint(0, 50, (16,), generator=torch.Generator().manual_seed(6))
    res_ = {}
    for emu in (False, True):
        sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
        emb_ref = effnet_ref.cv_predict_emb(sdr, name, x, use_fc=use_fc, training=True, emulate="fp16" if emu else None)
        loss_ref = arcface_ref.ce_loss(arcface_ref.arcface_forward(emb_ref, sdr["classifier.weight"], y, 64.0, 0.2), y)
        loss_ref.backward()
        res_[emu] = (emb_ref.detach(), loss_ref.item(), {k: v.grad for k, v in sdr.items() if torch.is_tensor(v) and v.grad is not None})
    from multimodalsimilar_amd import ops
    ops.set_deterministic(True)          # one adder per sum: the comparison below is of this build, not of one draw of atomic orderings
    try:
        model.to(DEV).train()
        emb = model.predict_emb(x.to(DEV))
        loss, _ = model.forward_loss(x.to(DEV), y.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ops.set_deterministic(False)
    emb_ref, loss_ref, grads = res_[False]
    emb_emu, loss_emu, grads_emu = res_[True]
    named = dict(model.named_parameters())
    gmax = max(v.norm().item() for v in grads.values())
    keys = [k for k in named if k in grads and named[k].grad is not None and grads[k].norm().item() > 1e-4 * gmax]
    ge = sorted(l2err(named[k].grad, grads[k]) for k in keys)
    g0 = sorted(l2err(grads_emu[k], grads[k]) for k in keys)
    e, d0 = l2err(emb, emb_ref), l2err(emb_emu, emb_ref)
    print(f"\n[{name} train, conditioned] emb L2 err {e:.4f} (emulation {d0:.4f}); loss {loss.item():.4f} vs {loss_ref:.4f} "
          f"(emulation {loss_emu:.4f}); grad L2 median {ge[len(ge) // 2]:.3f} / p90 {ge[int(0.9 * len(ge))]:.3f} "
          f"(emulation {g0[len(g0) // 2]:.3f} / {g0[int(0.9 * len(g0))]:.3f}) over {len(keys)} tensors")
    tag = f"image_tower_train_mode_conditioned[{name}]"
    record(tag, "oracle under fp16-storage emulation: embedding relative L2 (d0)", d0, NORTH_STAR)
    check(tag, "embedding relative L2", e, NORTH_STAR)
    check(tag, "loss relative error", abs(loss.item() - loss_ref) / loss_ref, NORTH_STAR)          # what the step optimises
    # gradients: bf16 gradient tensors through ~50-100 layers (the forward is fp16, the backward keeps bf16's range); over different
    # atomic orderings of the default mode the median moved by +-0.01 on B0, hence the deterministic run above
    check(tag, "median parameter-gradient relative L2", ge[len(ge) // 2], 3e-2)
    check(tag, "p90 parameter-gradient relative L2", ge[int(0.9 * len(ge))], 4e-2)


def test_two_tower_eval_embedding_and_forward_test_match_the_oracle():
    """MultimodalClassifier.predict_emb / forward(is_test=True) in eval mode (multimodal_classifier.py:27-57) against the oracle:
    conditioned EfficientNet-B0 + a 2-layer text tower, glue = normalise + concatenate, head = cosines."""
    from oracle import effnet_ref, bert_ref, arcface_ref
    from multimodalsimilar_amd.bert import BertModel, BertConfig
    from nlp_classifier import NlpClassifier
    from multimodal_classifier import MultimodalClassifier
    cv, sd_cv, _ = conditioned_cv("efficientnet_b0", False, seed=7, res=64, batch=12)
    shape = bert_ref.BertShape(512, 128, 2, 2, 512, 64)
    tsd = bert_ref.init_state(shape, seed=8)
    ptm = BertModel(BertConfig(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                               max_position_embeddings=64))
    ptm.load_state_dict(tsd)
    nlp = NlpClassifier(ptm, num_labels=40)
    model = MultimodalClassifier(DEV, cv, nlp, emb_size=1280 + 128, num_labels=40)
    model.eval()
    g = torch.Generator().manual_seed(10)
    B, S = 10, 32
    img = torch.randn(B, 3, 64, 64, generator=g)
    ids = torch.randint(0, 512, (B, S), generator=g)
    mask = (torch.arange(S).unsqueeze(0) < torch.randint(4, S + 1, (B, 1), generator=g)).long()
    with torch.no_grad():
        emb = model.predict_emb(img.to(DEV), ids.to(DEV), None, None, mask.to(DEV))
        cos = model(img.to(DEV), ids.to(DEV), None, None, mask.to(DEV), is_test=True)
    e_img = effnet_ref.cv_predict_emb(sd_cv, "efficientnet_b0", img, use_fc=False, training=False)
    e_txt = bert_ref.bert_forward(tsd, shape, ids, None, mask)
    ref = arcface_ref.glue_concat(e_img, e_txt)
    cos_ref = arcface_ref.arcface_forward_test(ref, model.classifier.weight.detach().cpu())
    e = l2err(emb, ref)
    e_img_emu = effnet_ref.cv_predict_emb(sd_cv, "efficientnet_b0", img, use_fc=False, training=False, emulate="fp16")
    d0 = l2err(torch.nn.functional.normalize(e_img_emu), torch.nn.functional.normalize(e_img))
    ei, et = l2err(emb[:, :1280], ref[:, :1280]), l2err(emb[:, 1280:], ref[:, 1280:])
    print(f"\n[two-tower eval] embedding L2 err {e:.4f}; halves: image {ei:.4f} (oracle under fp16-storage emulation {d0:.4f}), text {et:.4f}")
    tag = "two_tower_eval_embedding"
    record(tag, "oracle under fp16-storage emulation: image half relative L2 (d0)", d0, NORTH_STAR)
    check(tag, "text half relative L2", et, NORTH_STAR)                 # north_star: 1e-2 (bf16 text tower)
    check(tag, "image half relative L2", ei, NORTH_STAR)                # north_star: 1e-2 (fp16 image tower)
    check(tag, "whole embedding relative L2", e, NORTH_STAR)
    err = (cos.cpu() - cos_ref).abs().max()
    check(tag, "max |cos - cos_ref| (forward_test logits, unit scale)", err, NORTH_STAR)
    # predictions: wherever the reference's top-1 margin exceeds twice the measured cosine error the argmax must agree (random-weight
    # classes are near-tied: a fixed agreement rate over ten rows is a coin flip on the ties, not a property of the kernels)
    top2 = cos_ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * err
    assert (cos.argmax(1).cpu()[clear] == cos_ref.argmax(1)[clear]).all()
