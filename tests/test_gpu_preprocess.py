"""Input stage parity on the MI355X: multimodalsimilar_amd.preprocess (HIP, through the C ABI) against the
Pillow-pinned oracle and the committed Pillow fixture.  Integer work is bit-exact; ToTensor / Normalize are correctly rounded
fp32 operations on both sides, so the tensors are compared for equality."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fixture():
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess.npz")
    return {k: v for k, v in np.load(d).items()}


def test_matches_pillow_fixture_exactly():
    from multimodalsimilar_amd.preprocess import create_transform
    g = _fixture()
    for i, (H, W, S, cp) in enumerate(g["cases"]):
        tf = create_transform(input_size=(3, int(S), int(S)), interpolation="bicubic", crop_pct=float(cp))
        out = tf(g[f"img{i}"]).cpu().numpy()
        assert out.shape == (3, int(S), int(S))
        assert np.array_equal(out, g[f"out{i}"]), f"case {i}: max diff {np.abs(out - g[f'out{i}']).max()}"


@pytest.mark.parametrize("H,W", [(400, 400), (500, 375), (333, 801), (1200, 900), (320, 320), (321, 320), (64, 2000)])
def test_reference_config_against_oracle(H, W):
    """The reference's own config (multimodal_infer.py:86-90: 320, bicubic, crop_pct 1.0) on photo-sized inputs, down- and
    up-scaling, identity axes and extreme aspect ratios."""
    from multimodalsimilar_amd.preprocess import create_transform
    from oracle import preprocess_ref as P
    rng = np.random.default_rng(H * 7 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    tf = create_transform(input_size=(3, 320, 320), interpolation="bicubic", mean=(0.485, 0.456, 0.406),
                          std=(0.229, 0.224, 0.225), crop_pct=1.0)
    got = tf(img).cpu().numpy()
    assert np.array_equal(got, P.eval_transform(img, 320, 1.0))


def test_batch_of_ragged_sizes_and_pil_input():
    from multimodalsimilar_amd.preprocess import create_transform
    from oracle import preprocess_ref as P
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(250, 300), (224, 224), (500, 230), (231, 640)]]
    tf = create_transform(input_size=224, crop_pct=0.875)
    out = tf.batch([Image.fromarray(imgs[0]), imgs[1], torch.from_numpy(imgs[2]), torch.from_numpy(imgs[3]).cuda()])
    assert out.shape == (4, 3, 224, 224) and out.is_cuda
    for i, im in enumerate(imgs):
        assert np.array_equal(out[i].cpu().numpy(), P.eval_transform(im, 224, 0.875))
    # idempotent on the cached kernels, and a constant image stays constant (the kernels sum to one)
    assert torch.equal(tf.batch(imgs), out)
    flat = tf(np.full((300, 500, 3), 200, np.uint8))
    for c in range(3):
        assert float(flat[c].min()) == float(flat[c].max())


def test_errors_before_launch():
    from multimodalsimilar_amd.preprocess import create_transform
    tf = create_transform(input_size=224, crop_pct=1.0)
    with pytest.raises(TypeError):
        tf(np.zeros((300, 300), np.uint8))
    with pytest.raises(TypeError):
        tf(np.zeros((300, 300, 3), np.float32))
    with pytest.raises(TypeError):
        tf.into(np.zeros((300, 300, 3), np.uint8), torch.empty(3, 200, 200, device="cuda"))


def test_feeds_the_image_tower():
    """uint8 images -> input stage -> CvClassifier.predict_emb, the order of multimodal_infer.py:127-131."""
    from multimodalsimilar_amd.preprocess import create_transform
    from multimodalsimilar_amd import train as T
    cfg = dict(T.CONFIGS["tiny"], kind="cv", res=64, fc_dim=32, use_fc=True)
    model = T.build_model(cfg, "cuda", seed=0).eval()
    rng = np.random.default_rng(1)
    tf = create_transform(input_size=64, crop_pct=1.0)
    x = tf.batch([rng.integers(0, 256, (90, 70, 3), dtype=np.uint8) for _ in range(4)])
    with torch.no_grad():
        e = model.predict_emb(x)
    assert e.shape[0] == 4 and torch.isfinite(e).all()
