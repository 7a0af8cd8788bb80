"""The image-tower oracle (oracle/effnet_ref.py, PARITY UNPINNED w.r.t. timm: timm is neither vendored, pinned nor installed)
against an INDEPENDENT EfficientNet implementation that is installed: transformers.models.efficientnet.  That model is the
TensorFlow-port variant (TF "same" padding on stride-2 convs, BatchNorm eps 1e-3), so the oracle runs in its
``variant(tf_same=True, bn_eps=1e-3)`` mode -- the same block definitions with those two numerical switches.  Weights are
copied HF -> timm names; eval mode (running statistics), fp32, B0 and B4: outputs agree to float rounding.  This does not pin
parity with timm; it catches a wrong block definition (op order, squeeze-excite placement / width, channel rounding,
repeats, the skip rule), which is what an oracle written from a published description can get wrong."""
import pytest
import torch


def _hf_to_timm(hf_sd, arch):
    """transformers EfficientNetModel state dict -> the timm key names the oracle (and the HIP tower) use, 'backbone.' prefix."""
    out = {}

    def bn(src, dst):
        for leaf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            out[f"backbone.{dst}.{leaf}"] = hf_sd[f"{src}.{leaf}"].clone()

    out["backbone.conv_stem.weight"] = hf_sd["embeddings.convolution.weight"].clone()
    bn("embeddings.batchnorm", "bn1")
    for i, b in enumerate(arch["blocks"]):
        h, n = f"encoder.blocks.{i}", "backbone." + b["name"]
        if b["type"] == "ir":
            out[n + ".conv_pw.weight"] = hf_sd[h + ".expansion.expand_conv.weight"].clone()
            bn(h + ".expansion.expand_bn", b["name"] + ".bn1")
            d_bn, p_conv, p_bn = ".bn2", ".conv_pwl.weight", ".bn3"
        else:
            d_bn, p_conv, p_bn = ".bn1", ".conv_pw.weight", ".bn2"
        out[n + ".conv_dw.weight"] = hf_sd[h + ".depthwise_conv.depthwise_conv.weight"].clone()
        bn(h + ".depthwise_conv.depthwise_norm", b["name"] + d_bn)
        for a, c in (("reduce", "conv_reduce"), ("expand", "conv_expand")):
            out[n + f".se.{c}.weight"] = hf_sd[h + f".squeeze_excite.{a}.weight"].clone()
            out[n + f".se.{c}.bias"] = hf_sd[h + f".squeeze_excite.{a}.bias"].clone()
        out[n + p_conv] = hf_sd[h + ".projection.project_conv.weight"].clone()
        bn(h + ".projection.project_bn", b["name"] + p_bn)
    out["backbone.conv_head.weight"] = hf_sd["encoder.top_conv.weight"].clone()
    bn("encoder.top_bn", "bn2")
    return out


@pytest.mark.parametrize("name,width,depth", [("efficientnet_b0", 1.0, 1.0), ("efficientnet_b4", 1.4, 1.8)])
def test_oracle_blocks_agree_with_transformers_efficientnet(name, width, depth):
    from transformers import EfficientNetConfig, EfficientNetModel
    from oracle import effnet_ref
    arch = effnet_ref.arch(name)
    cfg = EfficientNetConfig(width_coefficient=width, depth_coefficient=depth, image_size=64, hidden_dim=arch["head"],
                             dropout_rate=0.0, drop_connect_rate=0.0)
    torch.manual_seed(0)
    hf = EfficientNetModel(cfg).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():                       # non-trivial BatchNorm state and SE biases, so that every term is exercised
        for k, p in hf.state_dict().items():
            if k.endswith("running_mean"):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
            elif k.endswith("running_var"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif ("norm" in k or "_bn" in k or "batchnorm" in k) and k.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif k.endswith("bias"):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
    hf_sd = hf.state_dict()
    assert len([k for k in hf_sd if k.startswith("encoder.blocks.") and k.endswith("depthwise_conv.depthwise_conv.weight")]) == len(arch["blocks"])
    sd = _hf_to_timm(hf_sd, arch)
    # the copied tensors have exactly the shapes the oracle's own initialiser gives them: same channel rounding / repeats / SE widths
    own = effnet_ref.init_state(name, seed=0)
    for k, v in own.items():
        if k.startswith("backbone."):
            assert k in sd and tuple(sd[k].shape) == tuple(v.shape), k
    assert set(k for k in own if k.startswith("backbone.")) == set(sd)
    x = torch.randn(3, 3, 64, 64, generator=g)
    with torch.no_grad():
        out = hf(pixel_values=x)
        with effnet_ref.variant(tf_same=True, bn_eps=cfg.batch_norm_eps):
            taps = {}
            feat = effnet_ref.backbone_forward(sd, name, x, training=False, taps=taps)
        pooled = feat.mean((2, 3))
        hs = hf(pixel_values=x, output_hidden_states=True).hidden_states
    assert feat.shape == out.last_hidden_state.shape
    # every block output, the feature map and the pooled embedding (cv_classifier.py:49-50)
    assert torch.allclose(taps["stem"], hs[0], rtol=1e-4, atol=1e-5)
    for i, b in enumerate(arch["blocks"]):
        assert torch.allclose(taps[b["name"]], hs[i + 1], rtol=2e-4, atol=2e-5), b["name"]
    assert torch.allclose(feat, out.last_hidden_state, rtol=5e-4, atol=5e-5)
    assert torch.allclose(pooled, out.pooler_output.flatten(1), rtol=5e-4, atol=5e-5)
    # and the default (timm) variant differs from it only through the two documented switches: with symmetric padding the
    # stride-2 layers see shifted windows, so the outputs must NOT be equal -- the switch is live
    with torch.no_grad():
        feat_timm = effnet_ref.backbone_forward(sd, name, x, training=False)
    assert not torch.allclose(feat_timm, feat, rtol=1e-3, atol=1e-4)
