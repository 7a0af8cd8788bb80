"""Pin the CPU oracle against the fixtures produced by running the reference (tests/golden/gen_golden.py)."""
import glob
import os
import numpy as np
import pytest
import torch

from oracle import arcface_ref, bert_ref, effnet_ref, optim_ref

T = torch.from_numpy


def _load(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name)).items()}


@pytest.mark.parametrize("i", range(5))
def test_arcface_oracle_matches_reference(golden_dir, i):
    d = _load(golden_dir, f"arcface_{i}.npz")
    x, w, y = T(d["x"]).requires_grad_(True), T(d["weight"]).requires_grad_(True), T(d["label"])
    s, m, easy = float(d["s"]), float(d["m"]), bool(int(d["easy"]))
    z = arcface_ref.arcface_forward(x, w, y, s, m, easy)
    loss = arcface_ref.ce_loss(z, y)
    loss.backward()
    assert torch.allclose(z, T(d["logits"]), rtol=0, atol=1e-5)
    assert torch.allclose(arcface_ref.arcface_forward_test(x.detach(), w.detach()), T(d["logits_test"]), atol=1e-6)
    assert abs(loss.item() - float(d["loss"])) < 1e-5
    assert torch.allclose(x.grad, T(d["dx"]), rtol=1e-4, atol=1e-6)
    assert torch.allclose(w.grad, T(d["dw"]), rtol=1e-4, atol=1e-6)
    # the closed-form backward (what the HIP head implements) agrees too
    l2, z2, dx2, dw2, am = arcface_ref.arcface_ce_analytic(x.detach(), w.detach(), y, s, m, easy)
    assert torch.allclose(z2, T(d["logits"]), atol=2e-5)
    assert abs(l2.item() - float(d["loss"])) < 2e-5
    assert torch.allclose(dx2, T(d["dx"]), rtol=1e-3, atol=2e-6)
    assert torch.allclose(dw2, T(d["dw"]), rtol=1e-3, atol=2e-6)
    assert torch.equal(am, T(d["logits"]).argmax(1))


def test_update_m_matches_reference(golden_dir):
    d = _load(golden_dir, "arcface_update_m.npz")
    m = 0.2
    for delta, row in zip(d["deltas"], d["traj"]):
        m = arcface_ref.update_m(m, float(delta))
        k = arcface_ref.margin_constants(m)
        assert np.allclose([m, k["cos_m"], k["sin_m"], k["th"], k["mm"]], row, atol=1e-12)


def test_glue_matches_reference(golden_dir):
    d = _load(golden_dir, "glue_0.npz")
    img, txt = T(d["img"]).requires_grad_(True), T(d["txt"]).requires_grad_(True)
    w = T(d["weight"]).requires_grad_(True)
    final = arcface_ref.glue_concat(img, txt)
    z = arcface_ref.arcface_forward(final, w, T(d["label"]), 64.0, 0.5, False)
    loss = arcface_ref.ce_loss(z, T(d["label"]))
    loss.backward()
    assert torch.allclose(final, T(d["final"]), atol=1e-6)
    assert torch.allclose(z, T(d["logits"]), atol=1e-5)
    assert torch.allclose(img.grad, T(d["dimg"]), rtol=1e-4, atol=1e-6)
    assert torch.allclose(txt.grad, T(d["dtxt"]), rtol=1e-4, atol=1e-6)
    assert torch.allclose(w.grad, T(d["dw"]), rtol=1e-4, atol=1e-6)


def _nlp_shape(name):
    if name == "tiny":
        return bert_ref.BertShape(128, 128, 2, 2, 512, 64)
    if name == "base1":      # BASELINE config 1's roberta-base shape, one layer
        return bert_ref.BertShape(21128, 768, 1, 12, 3072, 512)
    return bert_ref.BertShape(256, 256, 2, 4, 1024, 128)


def head_weight_from_golden(d, shape):
    """The ArcFace head of an nlp fixture: stored, or (base1: 3 MB) re-seeded exactly as the generator seeded it."""
    if "head_weight" in d:
        return T(d["head_weight"]).clone()
    C = d["logits"].shape[1]
    return torch.randn(C, shape.hidden_size, generator=torch.Generator().manual_seed(int(d["seed_head"]))) * 0.05


def nlp_state_from_golden(d, name):
    """Rebuild the BERT state the generator used (stored for 'tiny', re-seeded for 'mid')."""
    shape = _nlp_shape(name)
    if any(k.startswith("w::") for k in d):
        return shape, {k[3:]: T(v) for k, v in d.items() if k.startswith("w::")}
    sd = bert_ref.init_state(shape, seed=int(d["seed_state"]))
    g = torch.Generator().manual_seed(int(d["seed_perturb"]))
    for k in sd:
        if k.endswith("LayerNorm.weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith(".bias"):
            sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
    return shape, sd


@pytest.mark.parametrize("name", ["tiny", "mid", "base1"])
def test_bert_oracle_matches_reference(golden_dir, name):
    d = _load(golden_dir, f"nlp_{name}.npz")
    shape, sd = nlp_state_from_golden(d, name)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hw = head_weight_from_golden(d, shape).requires_grad_(True)
    pooled = bert_ref.bert_forward(sd, shape, T(d["input_ids"]), T(d["token_type_ids"]), T(d["attention_mask"]))
    assert torch.allclose(pooled, T(d["pooled"]), atol=2e-5)
    z = arcface_ref.arcface_forward(pooled, hw, T(d["label"]), 64.0, 0.40, False)   # nlp_classifier.py:15 defaults
    assert torch.allclose(z, T(d["logits"]), atol=2e-3)
    loss = arcface_ref.ce_loss(z, T(d["label"]))
    assert abs(loss.item() - float(d["loss"])) < 1e-3
    loss.backward()
    assert torch.allclose(hw.grad, T(d["head_grad"]).float(), rtol=1e-2, atol=1e-5)
    for k, v in d.items():
        if k.startswith("g::") or k.startswith("g64::"):
            g = sd[k.split("::")[1]].grad
            ref = T(v).float()                                   # base1 stores its 768 x 768 gradients as float16
            if k.startswith("g64::"):
                assert float(g[64:].abs().max()) == 0.0, k       # position rows >= S are never indexed
                g = g[:64]
            assert (g - ref).abs().max() <= 2e-3 * ref.abs().max() + 1e-6, k
        if k.startswith("gnorm::"):
            assert abs(sd[k[7:]].grad.norm().item() - float(v)) <= 2e-3 * float(v) + 1e-6, k


def test_multilabel_oracle_matches_reference(golden_dir):
    """nlp_classifier_multilabel.py: one embedding, three ArcFace heads (m 0.4 / 0.2 / 0.1), weighted CE sum;
    text-tower weights are nlp_tiny.npz's (same generator seeds)."""
    d = _load(golden_dir, "nlp_multilabel.npz")
    shape, sd = nlp_state_from_golden(_load(golden_dir, "nlp_tiny.npz"), "tiny")
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pooled = bert_ref.bert_forward(sd, shape, T(d["input_ids"]), T(d["token_type_ids"]), T(d["attention_mask"]))
    assert torch.allclose(pooled, T(d["pooled"]), atol=2e-5)
    heads = [T(d[f"head{i}"]).requires_grad_(True) for i in range(3)]
    loss = 0.0
    for i, m in enumerate((0.4, 0.2, 0.1)):
        z = arcface_ref.arcface_forward(pooled, heads[i], T(d[f"label{i}"]), 64.0, m, False)
        assert torch.allclose(z, T(d[f"logits{i}"]), atol=2e-3)
        assert torch.allclose(arcface_ref.arcface_forward_test(pooled, heads[i]), T(d[f"logits_test{i}"]), atol=1e-5)
        loss = loss + float(d["weights"][i]) * arcface_ref.ce_loss(z, T(d[f"label{i}"]))
    assert abs(loss.item() - float(d["loss"])) < 2e-3
    loss.backward()
    for i in range(3):
        assert torch.allclose(heads[i].grad, T(d[f"head_grad{i}"]), rtol=1e-2, atol=1e-5)
    for k, v in d.items():
        if k.startswith("g::"):
            ref = T(v)
            assert (sd[k[3:]].grad - ref).abs().max() <= 2e-3 * ref.abs().max() + 1e-6, k
        if k.startswith("gnorm::"):
            assert abs(sd[k[7:]].grad.norm().item() - float(v)) <= 2e-3 * float(v) + 1e-6, k


@pytest.mark.parametrize("tag", ["emb", "fc"])
def test_adamw_linear_schedule_matches_reference(golden_dir, tag):
    d = _load(golden_dir, f"adamw_{tag}.npz")
    lr0, warm, total = float(d["lr0"]), float(d["warmup"]), int(d["total"])
    ps = [T(d[f"p0_{j}"]).clone() for j in range(3)]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    for t in range(total):
        lr = optim_ref.linear_lr(lr0, t, warm, total)
        assert abs(lr - d["lrs"][t]) < 1e-12
        for j in range(3):
            ps[j], ms[j], vs[j] = optim_ref.adamw_step(ps[j], T(d[f"g{t}_{j}"]), ms[j], vs[j], t + 1, lr)
            assert torch.allclose(ps[j], T(d[f"p{t + 1}_{j}"]), rtol=1e-5, atol=1e-7)


def test_effnet_oracle_shapes_and_published_counts():
    # PARITY UNPINNED (no timm here); checks architecture against timm's published MAC/param counts.
    m0, p0 = effnet_ref.count_macs_params("efficientnet_b0")
    m4, p4 = effnet_ref.count_macs_params("efficientnet_b4")
    assert abs(m0 / 1e9 - 0.385) < 0.002 and abs(p0 / 1e6 - 4.01) < 0.01
    assert abs(m4 / 1e9 - 1.50) < 0.01 and abs(p4 / 1e6 - 17.55) < 0.01
    sd = effnet_ref.init_state("efficientnet_b0", fc_dim=16, seed=1)
    nparam = sum(v.numel() for k, v in sd.items() if k.startswith("backbone.") and
                 not any(s in k for s in ("running_", "num_batches")))
    assert nparam == p0
    x = torch.randn(2, 3, 64, 64)
    f = effnet_ref.backbone_forward(sd, "efficientnet_b0", x, training=True)
    assert f.shape == (2, 1280, 2, 2)
    e = effnet_ref.cv_predict_emb(sd, "efficientnet_b0", x, use_fc=True, training=True)
    assert e.shape == (2, 16) and torch.isfinite(e).all()
    a4 = effnet_ref.arch("efficientnet_b4")
    assert a4["stem"] == 48 and a4["head"] == 1792 and len(a4["blocks"]) == 32
    assert [b["cout"] for b in a4["blocks"]][:7] == [24, 24, 32, 32, 32, 32, 56]


def test_search_oracle_properties():
    """oracle/search_ref.py (faiss IndexFlat inner product, restated; unpinned): rows normalised, scores descending, self first,
    ties by ascending index, -1 / -inf padding when fewer than k vectors are stored."""
    import numpy as np
    from oracle import search_ref
    rng = np.random.default_rng(0)
    x = rng.standard_normal((40, 16)).astype(np.float32)
    x[11] = x[3]
    D, I = search_ref.search_inner_product(x, x, 5)
    assert np.all(np.diff(D, axis=1) <= 1e-7) and np.allclose(D[:, 0], 1.0, atol=1e-5)
    assert list(I[3, :2]) == [3, 11] and list(I[11, :2]) == [3, 11]
    D2, I2 = search_ref.search_inner_product(x[:2], x[:3], 5)
    assert np.all(I2[:, 3:] == -1) and np.all(np.isinf(D2[:, 3:]))
    assert np.allclose(np.linalg.norm(search_ref.normalize_l2(x), axis=1), 1.0, atol=1e-6)


def test_preprocess_oracle_matches_pillow_fixture(golden_dir):
    """Input stage oracle vs the fixture produced by Pillow + the torchvision steps (gen_golden.py gen_preprocess): the
    resized uint8 images bit for bit, the normalised tensors exactly."""
    from oracle import preprocess_ref as P
    g = _load(golden_dir, "preprocess.npz")
    for i, (H, W, S, cp) in enumerate(g["cases"]):
        img, S = g[f"img{i}"], int(S)
        oh, ow = g[f"resized{i}"].shape[:2]
        assert P.resize_target(int(H), int(W), int(np.floor(S / cp))) == (ow, oh)
        assert np.array_equal(P.resize_bicubic_u8(img, ow, oh), g[f"resized{i}"])
        assert np.array_equal(P.eval_transform(img, S, cp), g[f"out{i}"])


def test_preprocess_oracle_matches_installed_pillow():
    """The same pin live against the Pillow in this image, on shapes the fixture does not hold (up- and down-scaling,
    identity axis, extreme ratios)."""
    Image = pytest.importorskip("PIL.Image")
    from oracle import preprocess_ref as P
    rng = np.random.default_rng(3)
    for H, W, ow, oh in [(33, 47, 20, 14), (48, 36, 96, 128), (90, 60, 60, 90), (23, 23, 64, 64), (200, 150, 75, 56), (31, 17, 17, 31)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        assert np.array_equal(P.resize_bicubic_u8(img, ow, oh), np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC)))


def test_multimodal_forward_oracle_matches_the_reference_module(golden_dir):
    """multimodal_forward.npz was produced by EXECUTING /root/reference/multimodal_classifier.py:14-57 (tests/golden/gen_golden.py,
    gen_multimodal_forward: __init__ over two pickled towers, predict_emb, forward, forward(is_test=True)).  The oracle's glue +
    ArcFace(m = 0.5, s = 64) + text tower, around the stand-in image tower restated from its stored weights, must reproduce it."""
    d = _load(golden_dir, "multimodal_forward.npz")
    shape, sd = nlp_state_from_golden(d, "tiny")
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    fw, fb = T(d["cv_fc_weight"]).requires_grad_(True), T(d["cv_fc_bias"]).requires_grad_(True)
    hw = T(d["head_weight"]).requires_grad_(True)
    e_img = 3.0 * torch.tanh(torch.nn.functional.linear(T(d["img"]).mean((2, 3)), fw, fb))          # the generator's StandInImageTower
    e_txt = bert_ref.bert_forward(sd, shape, T(d["input_ids"]), T(d["token_type_ids"]), T(d["attention_mask"]))
    final = arcface_ref.glue_concat(e_img, e_txt)                                                  # multimodal_classifier.py:54-56
    assert torch.allclose(final, T(d["final"]), atol=2e-5)
    z = arcface_ref.arcface_forward(final, hw, T(d["label"]), 64.0, 0.5, False)                    # :22, :40
    assert torch.allclose(z, T(d["logits"]), atol=2e-3)
    assert torch.allclose(arcface_ref.arcface_forward_test(final, hw), T(d["logits_test"]), atol=2e-5)   # :42
    loss = arcface_ref.ce_loss(z, T(d["label"]))
    assert abs(loss.item() - float(d["loss"])) < 1e-3
    loss.backward()
    assert torch.allclose(hw.grad, T(d["head_grad"]), rtol=1e-2, atol=1e-5)
    assert torch.allclose(fw.grad, T(d["cv_fc_weight_grad"]), rtol=1e-2, atol=1e-5)
    assert torch.allclose(fb.grad, T(d["cv_fc_bias_grad"]), rtol=1e-2, atol=1e-5)
    for k in ("pooler.dense.weight", "encoder.layer.1.output.dense.weight"):
        ref = T(d["g::" + k])
        assert (sd[k].grad - ref).abs().max() <= 2e-3 * ref.abs().max() + 1e-6, k
