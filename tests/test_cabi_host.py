"""CPU-side checks (no GPU): the C-ABI library builds/loads and exports every symbol include/mmsim_hip.h declares;
host logic (flat buffers, HF state-dict names, whole-module pickles, loud failure without a GPU)."""
import ctypes
import io
import os
import pytest
import torch


def test_library_exports_every_declared_symbol():
    from multimodalsimilar_amd import build
    from multimodalsimilar_amd._lib import parse_header, LIBPATH, lib
    build.build(verbose=False)
    decls = parse_header()
    assert len(decls) >= 20
    dll = ctypes.CDLL(LIBPATH)
    for name in decls:
        assert hasattr(dll, name), name
    lib.load()
    assert lib.version() >= 300


def test_validation_happens_before_launch_and_reports_message():
    from multimodalsimilar_amd._lib import lib, MmsimError
    lib.load()
    with pytest.raises(MmsimError) as e:      # null operands: rejected by argument checks, nothing launched
        lib.gemm_bf16(0, 1, 16, 16, 16, None, 16, None, 16, None, 16, 1, None, 0, None, None, 0, 1.0, 1, 0, None)
    assert "gemm" in str(e.value)
    with pytest.raises(MmsimError):
        lib.attn_fwd(None, 0, None, None, 0, None, 1, 48, 2, 128, 0.0, 0, 0, None)   # S=48 unsupported


def _tiny():
    from multimodalsimilar_amd.bert import BertModel, BertConfig
    cfg = BertConfig(vocab_size=64, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=256, max_position_embeddings=32)
    return BertModel(cfg, seed=0)


def test_flat_buffer_views_and_hf_names():
    ptm = _tiny()
    sd = ptm.state_dict()
    for k in ("embeddings.word_embeddings.weight", "embeddings.LayerNorm.bias",
              "encoder.layer.1.attention.self.query.weight", "encoder.layer.0.attention.output.LayerNorm.weight",
              "encoder.layer.1.intermediate.dense.bias", "encoder.layer.0.output.dense.weight", "pooler.dense.bias"):
        assert k in sd, k
    fl = ptm._flat
    # parameters are views of the one flat buffer; q|k|v are adjacent (fused QKV view)
    for n, p in ptm.named_parameters():
        assert p.untyped_storage().data_ptr() == fl.master.untyped_storage().data_ptr()
    q, k, v = (f"encoder.layer.0.attention.self.{n}.weight" for n in ("query", "key", "value"))
    assert fl.offsets[k] - fl.offsets[q] == 128 * 128 and fl.offsets[v] - fl.offsets[k] == 128 * 128
    fused = fl._view(fl.master, q, (3 * 128, 128))
    assert torch.equal(fused[128:256], sd[k])
    assert all(o % 8 == 0 for o in fl.offsets.values()) and fl.total % 8 == 0
    # loading an HF-shaped state dict round-trips
    sd2 = {k_: torch.randn_like(v_) for k_, v_ in sd.items()}
    ptm.load_state_dict(sd2)
    assert torch.equal(ptm.state_dict()["pooler.dense.weight"], sd2["pooler.dense.weight"])


def test_nlp_classifier_dropin_surface_and_pickle():
    from nlp_classifier import NlpClassifier
    from arcface import ArcMarginProduct
    m = NlpClassifier(_tiny(), num_labels=10)
    assert isinstance(m.classifier, ArcMarginProduct) and m.classifier.in_feature == 128
    assert m.classifier.s == 64.0 and abs(m.classifier.m - 0.40) < 1e-12 and not m.classifier.easy_margin
    keys = set(m.state_dict().keys())
    assert "ptm.pooler.dense.weight" in keys and "emb_layer.ptm.pooler.dense.weight" in keys   # registered twice (H7)
    assert {"emb_layer.emb_layer.weight", "emb_layer.bn_layer.running_mean", "classifier.weight"} <= keys
    n_unique = sum(p.numel() for p in m.parameters())
    assert n_unique == m.ptm._flat.total - _pad(m.ptm._flat) + 10 * 128 + 128 * 128 + 128 + 2 * 128
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert type(m2).__module__ == "nlp_classifier"
    assert torch.equal(m2.classifier.weight, m.classifier.weight)
    assert m2.ptm._flat.master.untyped_storage().data_ptr() == m2.ptm.pooler.dense.bias.untyped_storage().data_ptr()
    # margin annealing (arcface.py:35-42)
    m.classifier.update_m(0.04)
    assert abs(m.classifier.m - 0.44) < 1e-12
    m.classifier.update_m(5.0)
    assert abs(m.classifier.m - 0.44) < 1e-12


def test_multilabel_classifier_dropin_surface():
    """nlp_classifier_multilabel.py:6-53: attributes, margins, state-dict keys, pickle by module path."""
    from nlp_classifier_multilabel import NlpClassifierMultilabel
    from arcface import ArcMarginProduct
    m = NlpClassifierMultilabel(_tiny(), 5, 7, 11)
    heads = (m.firstcate_classifier, m.secondcate_classifier, m.tag_classifier)
    assert all(isinstance(h, ArcMarginProduct) and h.in_feature == 128 and h.s == 64.0 for h in heads)
    assert [h.out_feature for h in heads] == [5, 7, 11] and [round(h.m, 6) for h in heads] == [0.4, 0.2, 0.1]
    keys = set(m.state_dict().keys())
    assert {"firstcate_classifier.weight", "secondcate_classifier.weight", "tag_classifier.weight",
            "ptm.pooler.dense.weight", "emb_layer.ptm.pooler.dense.weight"} <= keys
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert type(m2).__module__ == "nlp_classifier_multilabel"
    assert torch.equal(m2.tag_classifier.weight, m.tag_classifier.weight)
    if not torch.cuda.is_available():
        from multimodalsimilar_amd import MmsimError
        with pytest.raises(MmsimError):
            m(torch.zeros(2, 32, dtype=torch.long), firstcate_label=torch.zeros(2, dtype=torch.long),
              secondcate_label=torch.zeros(2, dtype=torch.long), tag_label=torch.zeros(2, dtype=torch.long))


def _pad(fl):
    used = 0
    for n in fl.names:
        k = 1
        for d in fl.shapes[n]:
            k *= d
        used += k
    return fl.total - used


def test_product_path_fails_loudly_without_gpu():
    from nlp_classifier import NlpClassifier
    from multimodalsimilar_amd import MmsimError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = NlpClassifier(_tiny(), num_labels=10)
    with pytest.raises(MmsimError):
        m(torch.zeros(2, 32, dtype=torch.long), label=torch.zeros(2, dtype=torch.long))
    with pytest.raises(MmsimError):
        m.classifier(torch.randn(2, 128), torch.zeros(2, dtype=torch.long))


def test_no_product_module_imports_the_oracle():
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, f) for f in os.listdir(root) if f.endswith(".py") and f not in ("bench.py", "__graft_entry__.py")]
    pkg = os.path.join(root, "multimodalsimilar_amd")
    files += [os.path.join(pkg, f) for f in os.listdir(pkg) if f.endswith(".py")]
    for f in files:
        src = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_preprocess_host_kernels_equal_the_oracles_and_reject_bad_input():
    """Host half of the input stage: the fixed-point resampling kernels built vectorised in the product equal the oracle's
    scalar restatement of Pillow's precompute_coeffs (itself pinned to Pillow), and argument errors raise before any launch."""
    import numpy as np
    import pytest
    from multimodalsimilar_amd import preprocess as PP
    from oracle import preprocess_ref as PR
    for a, b in [(53, 20), (48, 320), (500, 320), (23, 224), (100, 100), (17, 31), (1024, 224), (7, 3)]:
        ks, bo, kk = PR.resample_coeffs(a, b)
        ks2, b2, k2 = PP.bicubic_kernels(a, b)
        assert ks == ks2 and np.array_equal(bo, b2) and np.array_equal(kk, k2)
    with pytest.raises(ValueError):
        PP.create_transform(input_size=(3, 320, 300))
    with pytest.raises(ValueError):
        PP.create_transform(interpolation="bilinear")
    with pytest.raises(NotImplementedError):
        PP.create_transform(is_training=True)
    t = PP.create_transform(input_size=(3, 320, 320), interpolation="bicubic", mean=(0.485, 0.456, 0.406),
                            std=(0.229, 0.224, 0.225), crop_pct=1.0, device="cpu")       # the reference's config, multimodal_infer.py:86-90
    assert t.size == 320 and t.scale_size == 320
    with pytest.raises(PP.MmsimError):
        t(np.zeros((400, 400, 3), np.uint8))          # no CPU path
