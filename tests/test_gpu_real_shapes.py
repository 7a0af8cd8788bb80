"""The BASELINE configurations at their REAL shapes on the MI355X (model-level tests had only run tiny shapes,
so the production kernels -- the 256x256 LDS-DMA GEMM, the grouped weight gradient, the head at 100 000 / 1 000 000 classes,
EfficientNet-B4 at 224^2 x 256 -- were never selected in a parity test).

  cfg3 / cfg4 text tower   ONE BERT-large layer (H 1024, 16 heads, FFN 4096, vocab 21128) at S = 128, B = 256: forward +
                           backward against the CPU oracle (oracle/bert_ref.py), ~2.5 TFLOP of CPU work
  cfg4 head                B = 256, D = 2816, C = 100 000 against oracle/arcface_ref.py: loss, argmax, dX, sampled rows of dW
  cfg5 head                C = 1 000 000: size-independent properties (finite, the analytic initial loss, softmax-gradient rows
                           summing to zero, the label column carrying the only negative entry)
  cfg2 / cfg4 image tower  EfficientNet-B4 at 224 x 224, B = 256, one training step: finite loss, sane BatchNorm running
                           statistics, every gradient tensor non-zero and finite
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
from parity_log import check, record      # noqa: E402


def l2err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def test_bert_large_layer_at_cfg3_shape_matches_the_oracle():
    from oracle import bert_ref, arcface_ref
    from multimodalsimilar_amd import ops
    from multimodalsimilar_amd.bert import BertModel, BertConfig
    from nlp_classifier import NlpClassifier
    V, H, nh, I, S, B, C = 21128, 1024, 16, 4096, 128, 256, 10000
    shape = bert_ref.BertShape(vocab_size=V, hidden_size=H, num_hidden_layers=1, num_attention_heads=nh, intermediate_size=I,
                               max_position_embeddings=512)
    sd = bert_ref.init_state(shape, seed=11)
    cfg = BertConfig(vocab_size=V, hidden_size=H, num_hidden_layers=1, num_attention_heads=nh, intermediate_size=I,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ptm = BertModel(cfg)
    ptm.load_state_dict(sd)
    model = NlpClassifier(ptm, num_labels=C).to(DEV).train()
    # the production kernels must be the ones that run: pipelined 256x256 GEMM and the grouped weight gradient
    assert ops.wgrad_pair_eligible(3 * H, H, H, B * S)
    g = torch.Generator().manual_seed(12)
    ids = torch.randint(0, V, (B, S), generator=g)
    ids[:, 0] = 101
    mask = (torch.arange(S).unsqueeze(0) < torch.randint(8, S + 1, (B, 1), generator=g)).long()       # ragged, as padded batches are
    y = torch.randint(0, C, (B,), generator=g)
    loss, am = model.forward_loss(ids.to(DEV), None, None, mask.to(DEV), y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    hw = model.classifier.weight.detach().cpu().clone().requires_grad_(True)
    keys = ("pooler.dense.weight", "encoder.layer.0.output.dense.weight", "encoder.layer.0.intermediate.dense.weight",
            "encoder.layer.0.attention.self.query.weight", "encoder.layer.0.attention.self.value.weight",
            "encoder.layer.0.attention.output.dense.weight", "encoder.layer.0.intermediate.dense.bias",
            "encoder.layer.0.output.LayerNorm.weight", "encoder.layer.0.attention.output.dense.bias", "embeddings.LayerNorm.weight",
            "embeddings.position_embeddings.weight")
    sdr = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    torch.set_num_threads(16)
    pooled = bert_ref.bert_forward(sdr, shape, ids, None, mask)
    ref_logits = arcface_ref.arcface_forward(pooled, hw, y, 64.0, 0.40)
    ref_loss = arcface_ref.ce_loss(ref_logits, y)
    ref_loss.backward()
    tag = "bert_large_layer_cfg3_shape"
    check(tag, "loss relative error", abs(loss.item() - ref_loss.item()) / ref_loss.item(), 1e-2)
    with torch.no_grad():
        emb = model.predict_emb(ids.to(DEV), None, None, mask.to(DEV))
        logits = model(ids.to(DEV), None, None, mask.to(DEV), y.to(DEV))          # the literal path: materialised margin logits
    check(tag, "pooled embedding relative L2", l2err(emb, pooled), 1e-2)
    err = (logits.cpu() - ref_logits.detach()).abs().max().item()
    check(tag, "max |logit - oracle| / 64", err / 64.0, 1e-2)
    # predictions: 10 000 random classes leave the top two logits of many rows closer than the measured logit error -- on those
    # rows the argmax is a coin flip for ANY 16-bit path, not a property of the kernels.  Wherever the oracle's top-1 margin
    # exceeds twice the measured error the prediction must agree, for every such row; the raw agreement rate is recorded.
    top2 = ref_logits.detach().topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * err
    record(tag, "argmax agreement over all 256 rows (near-ties included)", (am.cpu() == ref_logits.argmax(1)).float().mean(), 2.0)
    record(tag, "rows whose oracle top-1 margin exceeds twice the logit error", clear.float().mean(), 2.0)
    assert clear.float().mean() > 0.5
    assert (am.cpu()[clear] == ref_logits.argmax(1)[clear]).all()
    named = dict(model.ptm.named_parameters())
    worst = max(l2err(named[k].grad, sdr[k].grad) for k in keys)
    check(tag, "worst parameter gradient relative L2 (11 tensors)", worst, 2e-2)
    check(tag, "head gradient relative L2", l2err(model.classifier.weight.grad, hw.grad), 1.5e-2)


def test_head_at_cfg4_shape_matches_the_oracle():
    from oracle import arcface_ref
    from multimodalsimilar_amd.head import ArcMarginProduct
    B, D, C = 256, 2816, 100000
    torch.manual_seed(0)
    head = ArcMarginProduct(D, C, m=0.5)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, D, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    # a few rows close to their class row: the margin branch with large cosines, where bf16 rounding matters most
    with torch.no_grad():
        x[:8] = head.weight[y[:8]] * 20 + 0.3 * x[:8]
    W = head.weight.detach().clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    torch.set_num_threads(16)
    logits = arcface_ref.arcface_forward(xr, W, y, 64.0, 0.5)
    ref = arcface_ref.ce_loss(logits, y)
    ref.backward()
    head.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    loss, am = head.forward_loss(xd, y.to(DEV))
    loss.backward()
    head.check_labels()
    tag = "head_cfg4_shape"
    check(tag, "loss relative error", abs(loss.item() - ref.item()) / ref.item(), 2e-3)
    record(tag, "argmax agreement over 256 rows", (am.cpu() == logits.argmax(1)).float().mean(), 2.0)
    assert (am.cpu() == logits.argmax(1)).float().mean() > 0.98
    check(tag, "dx relative L2", l2err(xd.grad, xr.grad), 1e-2)
    rows = torch.cat([y[:32], torch.randint(0, C, (64,), generator=g)])
    assert l2err(head.weight.grad.cpu()[rows], W.grad[rows]) < 2e-2
    check(tag, "dW relative L2", l2err(head.weight.grad, W.grad), 1e-2)
    cos = head.forward_test(xd.detach())
    check(tag, "max |forward_test cosine - oracle|", (cos.cpu() - arcface_ref.arcface_forward_test(x, W.detach())).abs().max(), 6e-3)
    # the literal API path at this shape: materialised margin logits
    lg = head(xd.detach(), y.to(DEV))
    assert (lg.cpu() - logits.detach()).abs().max() < 0.4          # 64-scale: 6e-3 on the cosines


def test_head_at_one_million_classes_properties():
    """cfg5's per-GPU head shape (replicated form).  No oracle at this size: properties that do not depend on it."""
    from multimodalsimilar_amd.head import ArcMarginProduct
    B, D, C, s, m = 256, 2816, 1000000, 64.0, 0.5
    head = ArcMarginProduct.__new__(ArcMarginProduct)
    torch.nn.Module.__init__(head)
    # construct on the device directly (xavier over [C, D] = 11.3 GB would take a minute on the host)
    from multimodalsimilar_amd.flat import FlatBuffer
    head.in_feature, head.out_feature, head.s, head.m, head.easy_margin = D, C, s, m, False
    head._flat = FlatBuffer([("weight", (C, D))], device=DEV)
    bound = math.sqrt(6.0 / (C + D))
    head._flat.master.uniform_(-bound, bound, generator=torch.Generator(device=DEV).manual_seed(0))
    head.weight = torch.nn.Parameter(head._flat.view("weight"))
    from multimodalsimilar_amd.head import _margin_consts
    head.cos_m, head.sin_m, head.th, head.mm = _margin_consts(m)
    head._scratch, head._wh_key, head._gen, head.grad_ready_hook = {}, None, 0, None
    x = torch.randn(B, D, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)).requires_grad_(True)
    y = torch.randint(0, C, (B,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    loss, am = head.forward_loss(x, y)
    loss.backward()
    head.check_labels()
    torch.cuda.synchronize()
    # analytic loss at random init: ln C + s sin m + s^2 / (2 D)  (cos ~ N(0, 1/D); the target logit is ~ -s sin m)
    expect = math.log(C) + s * math.sin(m) + s * s / (2 * D)
    assert math.isfinite(loss.item()) and abs(loss.item() - expect) < 0.02 * expect, (loss.item(), expect)
    assert torch.isfinite(x.grad).all() and torch.isfinite(head.weight.grad).all()
    assert 0 <= int(am.min()) and int(am.max()) < C
    # softmax-gradient rows sum to zero: dcos / (s slope) = (p - onehot) / B.  Off-target slope = 1, so row sums of dcos equal
    # -(1 - p_y)(slope_y - 1) s / B ... checked in the slope-free form: every entry except the label column is positive
    dcos = head._buf("dcos", (B, head._cpad()), torch.bfloat16)[:, :C].float()      # leading dimension: C rounded up to 256 (zero pad)
    neg = (dcos < 0).sum(1)
    assert int(neg.max()) == 1 and int(neg.min()) == 1
    assert torch.equal(dcos.argmin(1), y)
    off = dcos.clone()
    off[torch.arange(B), y] = 0
    p_off = off.sum(1) * B / s                                   # sum of the off-target probabilities
    assert ((p_off > 0.99) & (p_off < 1.001)).all()              # at init the target (margin-penalised) holds ~0 probability
    # and the weight gradient's rows are orthogonal to their weights (backward of the row normalisation)
    rows = torch.randint(0, C, (256,), device=DEV)
    gw, w = head.weight.grad[rows], head.weight.detach()[rows]
    assert ((gw * w).sum(1).abs() <= 2e-2 * gw.norm(dim=1) * w.norm(dim=1) + 1e-12).all()


def test_efficientnet_b4_at_224_batch_256_one_training_step():
    import warnings
    from cv_classifier import CvClassifier
    from multimodalsimilar_amd import train as T
    warnings.simplefilter("ignore")
    cfg = dict(kind="cv", image="efficientnet_b4", res=224, batch=256, classes=10000, fc_dim=512, use_fc=True)
    model = T.build_model(cfg, DEV, seed=0)
    loop = T.CvTrainLoop(model)
    batch = T.synthetic_batch(cfg, DEV, seed=5)
    w0 = model.backbone.conv_stem.weight.detach().clone()
    loss, pred = loop.step(batch)
    torch.cuda.synchronize()
    model.classifier.check_labels()
    D = 512
    expect = math.log(10000) + 64 * math.sin(0.2) + 64 * 64 / (2 * D)
    assert math.isfinite(loss.item()) and abs(loss.item() - expect) < 0.1 * expect, (loss.item(), expect)
    zero, total = [], 0
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        total += 1
        assert torch.isfinite(p.grad).all(), n
        if float(p.grad.abs().max()) == 0.0:
            zero.append(n)
    # optimizer.zero_grad() runs at the START of the next step: gradients of this step are still in the buffers
    assert total > 400 and not zero, zero[:5]
    bb = model.backbone
    for name in ("bn1", "blocks.0.0.bn1", "blocks.1.0.bn1", "blocks.1.0.bn2", "blocks.3.2.bn3", "blocks.5.7.bn2", "blocks.6.1.bn3", "bn2"):
        node = bb
        for part in name.split("."):
            node = getattr(node, part)
        rm, rv = node.running_mean, node.running_var
        assert torch.isfinite(rm).all() and torch.isfinite(rv).all() and (rv > 0).all(), name
        assert int(node.num_batches_tracked) == 1
        assert float((rv - 0.9).abs().max()) > 1e-4, name          # moved away from the initial 1.0 by the batch variance
    assert float((model.backbone.conv_stem.weight.detach() - w0).abs().max()) > 0      # Adam moved the first layer
