"""Host-side logic of the training step and the checkpoint loaders (CPU, no kernels): the order in which the two
optimisers and their schedules advance (multimodal_classifier_train.py:195-201), the timm-keyed state-dict loader
(cv_classifier.py:23 reads such a file through timm), and the HF BertModel -> native tower conversion
(nlp_classifier_train.py:63-64 hands NlpClassifier an HF BertModel)."""
import pytest
import torch


def _reference_lr_trajectory(steps, total, lr_emb=5e-5, lr_fc=1e-2, warm=0.15):
    """lr each optimiser USES at step t when driven exactly as the reference drives them: torch AdamW objects +
    transformers.get_scheduler('linear'), optimizer.step(); lr_scheduler.step() (multimodal_classifier_train.py:152-164,195-201)."""
    from transformers import get_scheduler
    pe, pf = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(3))
    oe, of = torch.optim.AdamW([pe], lr=lr_emb), torch.optim.AdamW([pf], lr=lr_fc)
    se = get_scheduler(name="linear", optimizer=oe, num_warmup_steps=0, num_training_steps=total)
    sf = get_scheduler(name="linear", optimizer=of, num_warmup_steps=warm * total, num_training_steps=total)
    used = []
    for _ in range(steps):
        pe.grad, pf.grad = torch.ones(3), torch.ones(3)
        used.append((oe.param_groups[0]["lr"], of.param_groups[0]["lr"]))
        oe.step(); se.step(); oe.zero_grad()
        of.step(); sf.step(); of.zero_grad()
    return used


def test_train_step_uses_lr_t_for_both_optimisers_like_the_reference():
    from multimodalsimilar_amd import train as T
    cfg = dict(kind="nlp", text="tiny", seq_len=32, batch=4, classes=16)
    model = T.build_model(cfg, "cpu", seed=0, dropout=False)
    total, steps = 20, 20
    ts = T.TrainStep(model, "nlp", total)
    used = []
    ts.opt_emb.step = lambda: used.append(["emb", ts.opt_emb.param_groups[0]["lr"]])
    ts.opt_fc.step = lambda: used.append(["fc", ts.opt_fc.param_groups[0]["lr"]])
    anchor = torch.zeros((), requires_grad=True)
    model.forward_loss = lambda **kw: (anchor * 1.0, torch.zeros(4, dtype=torch.long))
    batch = T.synthetic_batch(cfg, "cpu", seed=1)
    for _ in range(steps):
        ts.step(batch)
    ref = _reference_lr_trajectory(steps, total)
    got = [(used[2 * i][1], used[2 * i + 1][1]) for i in range(steps)]
    assert [u[0] for u in used[:2]] == ["emb", "fc"]
    for (ge, gf), (re_, rf) in zip(got, ref):
        assert abs(ge - re_) < 1e-15 and abs(gf - rf) < 1e-15
    assert got[0][1] == 0.0 and got[1][1] > 0.0          # the head's first warm-up step runs at lr 0, as in the reference
    assert got[-1][0] > 0.0                              # ... and the last step is not dropped (lr(T-1) > 0)


def test_oracle_step_advances_its_schedules_in_the_same_order():
    from oracle import step_ref
    orc = step_ref.TwoTowerOracle(None, {"w": torch.zeros(2, 2)}, None, None, torch.ones(4, 8), num_steps=20)
    ref = _reference_lr_trajectory(3, 20)
    for t in range(3):
        assert abs(orc.opt_emb.param_groups[0]["lr"] - ref[t][0]) < 1e-15
        assert abs(orc.opt_fc.param_groups[0]["lr"] - ref[t][1]) < 1e-15
        orc.t += 1
        orc._set_lr()


def test_timm_keyed_state_dict_loads_strictly():
    from multimodalsimilar_amd.effnet import EfficientNet
    from cv_classifier import load_timm_state_dict
    src = EfficientNet("efficientnet_b0", seed=3)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    for k in ("conv_stem.weight", "bn1.running_var", "blocks.0.0.conv_dw.weight", "blocks.0.0.se.conv_reduce.bias", "blocks.1.0.conv_pw.weight",
              "blocks.1.0.bn3.num_batches_tracked", "blocks.6.0.conv_pwl.weight", "conv_head.weight", "bn2.bias"):
        assert k in sd, k                                  # timm's own key names
    sd["classifier.weight"], sd["classifier.bias"] = torch.zeros(1000, 1280), torch.zeros(1000)      # ImageNet head the reference strips
    dst = EfficientNet("efficientnet_b0", seed=4)
    load_timm_state_dict(dst, sd)
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k
    assert torch.equal(dst._flat.view("conv_head.weight"), src._flat.view("conv_head.weight"))       # landed in the flat buffer
    bad = dict(sd); bad.pop("blocks.3.1.bn2.weight")
    with pytest.raises(KeyError):
        load_timm_state_dict(EfficientNet("efficientnet_b0"), bad)
    bad = dict(sd); bad["blocks.9.0.conv_pw.weight"] = torch.zeros(1)
    with pytest.raises(KeyError):
        load_timm_state_dict(EfficientNet("efficientnet_b0"), bad)
    bad = dict(sd); bad["conv_head.weight"] = torch.zeros(1280, 320)        # wrong rank
    with pytest.raises(KeyError):
        load_timm_state_dict(EfficientNet("efficientnet_b0"), bad)


def test_as_native_copies_every_weight_of_an_hf_bert_model():
    import transformers
    from multimodalsimilar_amd.bert import as_native
    cfg = transformers.BertConfig(vocab_size=97, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                  max_position_embeddings=40, type_vocab_size=2)
    torch.manual_seed(0)
    hf = transformers.BertModel(cfg)
    native = as_native(hf)
    hsd = {k: v for k, v in hf.state_dict().items() if "position_ids" not in k}
    nsd = native.state_dict()
    assert set(hsd) == set(nsd)
    for k, v in hsd.items():
        assert torch.equal(nsd[k], v), k
    assert native.config.num_attention_heads == 2 and native.config.layer_norm_eps == cfg.layer_norm_eps
    assert as_native(native) is native


def test_cv_loop_schedule_and_margin_annealing_follow_the_reference_objects():
    """The image-only loop's lr (per epoch) is torch's CosineAnnealingWarmRestarts(T_0=7, eta_min=1e-6) on Adam(lr=1e-3) and the
    margin grows by 0.04 per epoch, clamped as ArcMarginProduct.update_m clamps it (cv_classifier_train_daodian.py:264-267,292;
    arcface.py:35-42)."""
    import math
    from multimodalsimilar_amd import train as T
    cfg = dict(kind="cv", image="efficientnet_b0", res=64, batch=4, classes=32, fc_dim=64, use_fc=True)
    model = T.build_model(cfg, "cpu", seed=0)
    loop = T.CvTrainLoop(model)
    p = torch.nn.Parameter(torch.zeros(2))
    ref_opt = torch.optim.Adam([p], lr=1e-3)
    ref_sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(ref_opt, T_0=7, T_mult=1, eta_min=1e-6, last_epoch=-1)
    g = loop.optimizer.param_groups[0]
    assert g["weight_decay"] == 0.0 and g["betas"] == (0.9, 0.999) and g["eps"] == 1e-8
    # every parameter of the model is under the ONE optimiser (model.parameters() in the reference): backbone, fc/bn top, head
    flats = {id(f) for f in loop.optimizer.flats}
    assert id(model.backbone._flat) in flats and id(model._flat) in flats and id(model.classifier._flat) in flats
    m0 = model.classifier.m
    for epoch in range(16):
        assert abs(g["lr"] - ref_opt.param_groups[0]["lr"]) < 1e-12, epoch
        assert abs(model.classifier.m - min(m0 + 0.04 * epoch, 1.0)) < 1e-9 or model.classifier.m <= 1.0
        assert abs(model.classifier.cos_m - math.cos(model.classifier.m)) < 1e-12
        ref_opt.step(); ref_sched.step()
        loop.end_epoch()
    assert abs(g["lr"] - ref_opt.param_groups[0]["lr"]) < 1e-12 and loop.epoch == 16
    with pytest.raises(ValueError):
        from multimodalsimilar_amd.optim import FusedAdam
        FusedAdam(model, weight_decay=0.01)


def test_flat_buffer_lazy_zero_grad_flag():
    """flat.zero_grad(lazy=True): only a buffer whose owner declared a full-coverage overwriting writer is marked instead of
    filled; the writer takes the mark exactly once; a reader that needs zeros materialises them."""
    import torch
    from multimodalsimilar_amd.flat import FlatBuffer
    f = FlatBuffer([("weight", (4, 8))], device="cpu")
    f.grad = torch.ones(f.total)
    f.zero_grad(lazy=True)
    assert not f.zero_pending and float(f.grad.sum()) == 0          # not overwrite-capable: eager fill
    f.grad.fill_(1)
    f.overwrite_capable = True
    f.zero_grad(lazy=True)
    assert f.zero_pending and float(f.grad.sum()) == f.total        # untouched, marked
    assert f.take_zero_pending() and not f.take_zero_pending()
    f.zero_grad(lazy=True)
    f.materialize_zero()
    assert float(f.grad.sum()) == 0 and not f.zero_pending
    f.grad.fill_(1)
    f.zero_grad()                                                   # the default stays eager
    assert float(f.grad.sum()) == 0 and not f.zero_pending
