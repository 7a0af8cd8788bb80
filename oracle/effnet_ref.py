"""Oracle: EfficientNet image tower + CvClassifier top, fp32 CPU.  Test infrastructure only.

PARITY UNPINNED.  The reference builds the tower with ``timm.create_model(model_name)``
(cv_classifier.py:23-27); timm is not vendored in /root/reference, not pinned by it and not
installed in the build container, and the reference's tests hold no numeric vectors for it
(image_emb_test.py:42 prints a tensor).  This file restates timm's published
``efficientnet_b0`` / ``efficientnet_b4`` architecture: PyTorch-symmetric
padding k//2, BatchNorm eps 1e-5 momentum 0.1, SiLU, SE with rd = round(block_in * 0.25),
channel rounding make_divisible(c*w, 8, round_limit 0.9), repeats ceil(r*d), skip when
stride 1 and cin == cout.  It reproduces timm's published 0.385 GMAC (B0) / 1.50 GMAC (B4)
at 224x224 and 4.01 M / 17.55 M backbone parameters (tests/test_oracle_effnet.py).

Tower top follows cv_classifier.py:47-55: AdaptiveAvgPool2d(1) -> Dropout(0.5) -> Linear ->
BatchNorm1d.  Dropout is the identity here (parity runs use p = 0).
State-dict keys are timm's.
"""
import math
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# Numerical variant of the same architecture.  The default is timm's `efficientnet_b*` (PyTorch-symmetric padding k//2, BatchNorm
# eps 1e-5): what the reference builds.  variant(tf_same=True, bn_eps=1e-3) is the TensorFlow-port variant (stride-2 convs pad
# (k//2 - 1, k//2) = TF "same" on even inputs, eps 1e-3) that `transformers.models.efficientnet` implements -- an INDEPENDENT
# implementation that IS installed here, used by tests/test_oracle_effnet_hf.py to check this restatement's block definitions
# (op order, SE placement and width, channel rounding, repeats, skip rule).  It does not pin parity with timm (PARITY UNPINNED
# stays, see the header); it catches a wrong block.
_VARIANT = {"tf_same": False, "bn_eps": BN_EPS}


class variant:
    def __init__(self, tf_same=False, bn_eps=BN_EPS):
        self.new = {"tf_same": tf_same, "bn_eps": bn_eps}

    def __enter__(self):
        self.old = dict(_VARIANT)
        _VARIANT.update(self.new)

    def __exit__(self, *a):
        _VARIANT.update(self.old)


def _conv_k(x, w, stride, k, groups=1):
    """k x k conv, padding k//2 -- or, in the tf_same variant, TF "same" padding for stride 2 (right / bottom heavier)."""
    if _VARIANT["tf_same"] and stride == 2:
        x = F.pad(x, (k // 2 - 1, k // 2, k // 2 - 1, k // 2))
        return F.conv2d(x, w, None, stride=stride, padding=0, groups=groups)
    return F.conv2d(x, w, None, stride=stride, padding=k // 2, groups=groups)

# (type, repeats, kernel, stride, expand, out_channels) -- the B0 base, scaled by (width, depth)
_BASE = [("ds", 1, 3, 1, 1, 16), ("ir", 2, 3, 2, 6, 24), ("ir", 2, 5, 2, 6, 40), ("ir", 3, 3, 2, 6, 80),
         ("ir", 3, 5, 1, 6, 112), ("ir", 4, 5, 2, 6, 192), ("ir", 1, 3, 1, 6, 320)]
_SCALE = {"efficientnet_b0": (1.0, 1.0), "efficientnet_b4": (1.4, 1.8)}


def make_divisible(v, divisor=8, round_limit=0.9):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def arch(model_name):
    """Returns dict(stem, head, blocks=[dict(type,cin,mid,cout,k,stride,rd,skip,name)])."""
    w, d = _SCALE[model_name]
    stem = make_divisible(32 * w)
    head = make_divisible(1280 * w)
    blocks = []
    cin = stem
    for si, (typ, r, k, s, e, c) in enumerate(_BASE):
        cout = make_divisible(c * w)
        reps = int(math.ceil(r * d))
        for bi in range(reps):
            stride = s if bi == 0 else 1
            mid = cin * e
            blocks.append(dict(type=typ, cin=cin, mid=mid, cout=cout, k=k, stride=stride,
                               rd=int(round(cin * 0.25)), skip=(stride == 1 and cin == cout),
                               name=f"blocks.{si}.{bi}"))
            cin = cout
    return dict(stem=stem, head=head, blocks=blocks, last=cin)


def init_state(model_name, fc_dim=None, seed=0):
    """timm-style init: conv normal(0, sqrt(2/fan_out)), BN gamma 1 beta 0, SE conv bias 0."""
    g = torch.Generator().manual_seed(seed)
    a = arch(model_name)
    sd = {}

    def conv(name, cout, cin_g, k, groups=1, bias=False):
        fan_out = k * k * cout // groups
        sd[name + ".weight"] = torch.randn(cout, cin_g, k, k, generator=g) * math.sqrt(2.0 / fan_out)
        if bias:
            sd[name + ".bias"] = torch.zeros(cout)

    def bn(name, c):
        sd[name + ".weight"] = torch.ones(c)
        sd[name + ".bias"] = torch.zeros(c)
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    conv("conv_stem", a["stem"], 3, 3)
    bn("bn1", a["stem"])
    for b in a["blocks"]:
        n = b["name"]
        if b["type"] == "ds":
            conv(n + ".conv_dw", b["mid"], 1, b["k"], groups=b["mid"]); bn(n + ".bn1", b["mid"])
            conv(n + ".se.conv_reduce", b["rd"], b["mid"], 1, bias=True)
            conv(n + ".se.conv_expand", b["mid"], b["rd"], 1, bias=True)
            conv(n + ".conv_pw", b["cout"], b["mid"], 1); bn(n + ".bn2", b["cout"])
        else:
            conv(n + ".conv_pw", b["mid"], b["cin"], 1); bn(n + ".bn1", b["mid"])
            conv(n + ".conv_dw", b["mid"], 1, b["k"], groups=b["mid"]); bn(n + ".bn2", b["mid"])
            conv(n + ".se.conv_reduce", b["rd"], b["mid"], 1, bias=True)
            conv(n + ".se.conv_expand", b["mid"], b["rd"], 1, bias=True)
            conv(n + ".conv_pwl", b["cout"], b["mid"], 1); bn(n + ".bn3", b["cout"])
    conv("conv_head", a["head"], a["last"], 1)
    bn("bn2", a["head"])
    sd = {"backbone." + k: v for k, v in sd.items()}
    if fc_dim is not None:
        bound = 1.0 / math.sqrt(a["head"])
        sd["fc.weight"] = (torch.rand(fc_dim, a["head"], generator=g) * 2 - 1) * bound
        sd["fc.bias"] = (torch.rand(fc_dim, generator=g) * 2 - 1) * bound
        sd["bn.weight"] = torch.ones(fc_dim)
        sd["bn.bias"] = torch.zeros(fc_dim)
        sd["bn.running_mean"] = torch.zeros(fc_dim)
        sd["bn.running_var"] = torch.ones(fc_dim)
        sd["bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return sd


def _bn(x, sd, name, training, stats=None):
    if training:
        dims = [0] + list(range(2, x.dim()))
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        if stats is not None:
            stats[name] = (mean.detach().clone(), var.detach().clone())
    else:
        mean, var = sd[name + ".running_mean"], sd[name + ".running_var"]
    shp = [1, -1] + [1] * (x.dim() - 2)
    return (x - mean.view(shp)) / torch.sqrt(var.view(shp) + _VARIANT["bn_eps"]) * sd[name + ".weight"].view(shp) \
        + sd[name + ".bias"].view(shp)


def _se(x, sd, n):
    s = x.mean((2, 3), keepdim=True)
    s = F.silu(F.conv2d(s, sd[n + ".se.conv_reduce.weight"], sd[n + ".se.conv_reduce.bias"]))
    s = F.conv2d(s, sd[n + ".se.conv_expand.weight"], sd[n + ".se.conv_expand.bias"])
    return x * torch.sigmoid(s)


def _q(t, on):
    """Storage emulation (round to the storage type, compute in fp32); the gradient passes straight through the casts.
    on: False / None = off; True or "bf16" = bf16 (the rounds-1/2 layout of the MI355X path); "fp16" = IEEE half (the layout since
    round 3: forward tensors of the image tower are stored as fp16)."""
    if not on:
        return t
    return t.half().float() if on == "fp16" else t.bfloat16().float()


def mbconv_forward(sd, n, b, x, training=True, stats=None, e=False):
    """One DS / IR block; n = state-dict prefix of the block, b = its arch() entry."""
    sc = x
    if b["type"] == "ds":
        x = _q(_conv_k(x, sd[n + ".conv_dw.weight"], b["stride"], b["k"], groups=b["mid"]), e)
        x = F.silu(_bn(x, sd, n + ".bn1", training, stats))
        x = _q(_se(x, sd, n), e)
        x = _q(F.conv2d(x, _q(sd[n + ".conv_pw.weight"], e)), e)
        x = _bn(x, sd, n + ".bn2", training, stats)
    else:
        x = _q(F.conv2d(x, _q(sd[n + ".conv_pw.weight"], e)), e)
        x = _q(F.silu(_bn(x, sd, n + ".bn1", training, stats)), e)
        x = _q(_conv_k(x, sd[n + ".conv_dw.weight"], b["stride"], b["k"], groups=b["mid"]), e)
        x = F.silu(_bn(x, sd, n + ".bn2", training, stats))
        x = _q(_se(x, sd, n), e)
        x = _q(F.conv2d(x, _q(sd[n + ".conv_pwl.weight"], e)), e)
        x = _bn(x, sd, n + ".bn3", training, stats)
    if b["skip"]:
        x = x + sc
    return _q(x, e)


def backbone_forward(sd, model_name, x, training=True, stats=None, taps=None, emulate_bf16=False, emulate=None):
    """x [B,3,H,W] -> feature map [B,head,H/32,W/32].  sd keys carry the 'backbone.' prefix.

    emulate="fp16" / "bf16" (emulate_bf16=True == "bf16") rounds to that type exactly where the MI355X path stores 16-bit
    tensors (conv outputs before BatchNorm, activated tensors that are materialised, block outputs, 1x1-conv weights) and keeps
    fp32 everywhere else.  It is the same algorithm; it separates storage error from implementation error in the parity tests."""
    a = arch(model_name)
    p = "backbone."
    e = emulate if emulate else emulate_bf16
    x = _q(_conv_k(x, sd[p + "conv_stem.weight"], 2, 3), e)
    x = _q(F.silu(_bn(x, sd, p + "bn1", training, stats)), e)
    if taps is not None:
        taps["stem"] = x
    for b in a["blocks"]:
        x = mbconv_forward(sd, p + b["name"], b, x, training, stats, e)
        if taps is not None:
            taps[b["name"]] = x
    x = _q(F.conv2d(x, _q(sd[p + "conv_head.weight"], e)), e)
    x = F.silu(_bn(x, sd, p + "bn2", training, stats))
    return x


def cv_predict_emb(sd, model_name, x, use_fc=True, training=True, stats=None, taps=None, emulate_bf16=False, emulate=None):
    """cv_classifier.py:47-55 (dropout = identity)."""
    em = emulate if emulate else emulate_bf16
    f = backbone_forward(sd, model_name, x, training, stats, taps, emulate=em)
    e = f.mean((2, 3))                                    # AdaptiveAvgPool2d(1).view(B,-1)   :50
    if use_fc:
        e = F.linear(_q(e, em), _q(sd["fc.weight"], em), sd["fc.bias"])   # :53
        e = _bn(e, sd, "bn", training, stats)             # :54
    return e


def count_macs_params(model_name, res=224):
    a = arch(model_name)
    macs = 0
    params = 0
    h = res // 2
    macs += h * h * a["stem"] * 27; params += a["stem"] * 27 + 2 * a["stem"]
    for b in a["blocks"]:
        ho = h // b["stride"]
        if b["type"] == "ir":
            macs += h * h * b["cin"] * b["mid"]; params += b["cin"] * b["mid"] + 2 * b["mid"]
        macs += ho * ho * b["mid"] * b["k"] ** 2; params += b["mid"] * b["k"] ** 2 + 2 * b["mid"]
        macs += 2 * b["mid"] * b["rd"]; params += 2 * b["mid"] * b["rd"] + b["mid"] + b["rd"]
        macs += ho * ho * b["mid"] * b["cout"]; params += b["mid"] * b["cout"] + 2 * b["cout"]
        h = ho
    macs += h * h * a["last"] * a["head"]; params += a["last"] * a["head"] + 2 * a["head"]
    return macs, params
