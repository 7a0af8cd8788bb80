"""Drop-in for the reference's ``cv_classifier`` module (cv_classifier.py:17-55): EfficientNet image tower +
optional Dropout(0.5) -> Linear -> BatchNorm1d top + ArcFace(m=0.2) head, on the MI355X HIP path.

Same constructor / attributes (backbone, pooling, dropout, fc, bn, classifier, use_fc, num_labels) / ``forward`` /
``predict_emb`` as the reference.  ``pretrained=True`` cannot download here: weights are read from
``$MMSIM_PRETRAINED_DIR/<model_name>.pth`` (timm state-dict names) when present, else the tower is random-init.
"""
import os
import warnings

import torch
import torch.nn as nn

from arcface import ArcMarginProduct
from multimodalsimilar_amd import ops
from multimodalsimilar_amd.effnet import EfficientNet
from multimodalsimilar_amd.flat import FlatBuffer
from multimodalsimilar_amd._lib import lib, MmsimError

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def load_timm_state_dict(backbone, sd):
    """Load a timm-format EfficientNet state dict (what ``timm.create_model(..., pretrained=True)`` reads, cv_classifier.py:23;
    file ``efficientnet_b4_ra2_320-7eb33cd5.pth``, cv_classifier_train.py:27) into the HIP backbone.  timm checkpoints carry the
    ImageNet classifier the reference strips (``classifier.weight / .bias``, cv_classifier.py:24-27): those two keys are the ONLY
    ones that may be unexpected; any other missing / unexpected / mis-shaped key raises instead of being skipped silently."""
    sd = dict(sd.get("state_dict", sd)) if isinstance(sd, dict) else sd
    for k in ("classifier.weight", "classifier.bias"):
        sd.pop(k, None)
    own = backbone.state_dict()
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    shapes = [k for k in sd if k in own and tuple(sd[k].shape) != tuple(own[k].shape)]
    if missing or unexpected or shapes:
        raise KeyError(f"timm state dict does not match {backbone.model_name}: missing {missing[:5]} (+{max(0, len(missing) - 5)}), "
                       f"unexpected {unexpected[:5]} (+{max(0, len(unexpected) - 5)}), shape mismatch {shapes[:5]}")
    backbone.load_state_dict(sd, strict=True)


class CvClassifier(nn.Module):
    def __init__(self, model_name, fc_dim, num_labels, m=0.2, pretrained=True, use_fc=True):
        super().__init__()
        self.backbone = EfficientNet(model_name)
        if pretrained:
            path = os.path.join(os.environ.get("MMSIM_PRETRAINED_DIR", ""), model_name + ".pth")
            if os.path.isfile(path):
                load_timm_state_dict(self.backbone, torch.load(path, map_location="cpu"))
            else:
                warnings.warn(f"CvClassifier: no local weights for {model_name!r} (offline; set MMSIM_PRETRAINED_DIR) - "
                              "the image tower is randomly initialised")
        in_features = self.backbone.num_features
        self.pooling = nn.AdaptiveAvgPool2d(1)        # kept for the attribute contract; the pool is fused on the HIP path
        self.use_fc = use_fc
        self.num_labels = num_labels
        self._flat = None
        if self.use_fc:
            self.dropout = nn.Dropout(0.5)
            self.fc = nn.Linear(in_features, fc_dim)
            self.bn = nn.BatchNorm1d(fc_dim)
            self._flat = FlatBuffer([("fc.weight", (fc_dim, in_features)), ("fc.bias", (fc_dim,)),
                                     ("bn.weight", (fc_dim,)), ("bn.bias", (fc_dim,))], device="cpu", f16_shadow=True)
            with torch.no_grad():
                for n, p in self._top_params():
                    self._flat.view(n).copy_(p)
            self._rebind_top()
            in_features = fc_dim
        self.classifier = ArcMarginProduct(in_features, self.num_labels, m=m)
        self._step_seed = 0
        self.grad_ready_hook = None

    # ---- flat-buffer plumbing for the fc/bn top
    def _top_params(self):
        return [("fc.weight", self.fc.weight), ("fc.bias", self.fc.bias), ("bn.weight", self.bn.weight), ("bn.bias", self.bn.bias)]

    def _rebind_top(self):
        for n, p in self._top_params():
            p.data = self._flat.view(n)
            p.grad = None

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        if self._flat is not None:
            self._flat.apply_(fn)
            self._rebind_top()
        return self

    def flat_buffers(self):
        return [self._flat] if self._flat is not None else []

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        if self._flat is not None:
            self._flat._shadow_version = None

    def _bind_top_grads(self):
        self._flat.ensure_device_state()
        for n, p in self._top_params():
            g = self._flat.gview(n)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    # ---- reference API
    def forward(self, input: torch.Tensor, label=None, is_test=False):
        img_embedding = self.predict_emb(input)
        if not is_test:
            return self.classifier(img_embedding, label)
        return self.classifier.forward_test(img_embedding)

    def forward_loss(self, input: torch.Tensor, label):
        return self.classifier.forward_loss(self.predict_emb(input), label)

    def predict_emb(self, inp: torch.Tensor):
        pooled = self.backbone.forward_pooled(inp)                  # backbone -> pooling -> view   (:49-50)
        if not self.use_fc:
            return pooled
        return _CvTopFn.apply(pooled, self.fc.weight, self)         # dropout -> fc -> bn            (:52-54)


class _CvTopFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pooled, fc_weight, mod):
        if not pooled.is_cuda:
            raise MmsimError("CvClassifier: inputs must be on the GPU; the HIP path has no CPU fallback")
        fl = mod._flat
        fl.sync_shadow()
        s = ops._stream()
        B, Cin = pooled.shape
        fc_dim = mod.fc.out_features
        pooled = pooled.contiguous().float()
        train = mod.training
        p = mod.dropout.p if train else 0.0
        if train:
            mod._step_seed += 1
        seed = (mod._step_seed * 0x9E3779B97F4A7C15 + 0xC0FFEE) & 0xFFFFFFFFFFFFFFFF
        # the image tower's forward tensors are fp16 (effnet.py): the pooled features and the fc weight enter the MFMAs as fp16
        xb = torch.empty(B, Cin, dtype=torch.float16, device=pooled.device)
        lib.dropout_cast(pooled.data_ptr(), xb.data_ptr(), B * Cin, p, seed, 1, s)
        y = torch.empty(B, fc_dim, dtype=torch.float32, device=pooled.device)
        ops.gemm(xb, fl.sview16("fc.weight"), y, bias=fl.view("fc.bias"))
        out = torch.empty_like(y)
        mean = torch.empty(fc_dim, dtype=torch.float32, device=y.device)
        rstd = torch.empty_like(mean)
        lib.bn1d_fwd(y.data_ptr(), fl.view("bn.weight").data_ptr(), fl.view("bn.bias").data_ptr(), out.data_ptr(), mean.data_ptr(),
                     rstd.data_ptr(), mod.bn.running_mean.data_ptr(), mod.bn.running_var.data_ptr(), B, fc_dim, BN_EPS,
                     BN_MOMENTUM, int(train), s)
        if train:
            mod.bn.num_batches_tracked += 1
        ctx.mod, ctx.saved = mod, (xb, y, mean, rstd, p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        xb, y, mean, rstd, p, seed = ctx.saved
        fl = mod._flat
        mod._bind_top_grads()
        s = ops._stream()
        B, fc_dim = y.shape
        Cin = xb.shape[1]
        dout = dout.contiguous().float()
        dy = torch.empty_like(y)
        lib.bn1d_bwd(dout.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), fl.view("bn.weight").data_ptr(), dy.data_ptr(),
                     fl.gview("bn.weight").data_ptr(), fl.gview("bn.bias").data_ptr(), B, fc_dim, s)
        dyb = ops.alloc_2d(B, fc_dim, torch.bfloat16, dy.device, zero=True)
        dyb.copy_(dy)
        ops.gemm(dyb, xb, fl.gview("fc.weight"), trans_a=True, b_kmajor=False, accumulate=True)
        ops.colsum(dyb, fl.gview("fc.bias"))
        dxd = torch.empty(B, Cin, dtype=torch.float32, device=dy.device)
        ops.gemm(dyb, fl.sview("fc.weight"), dxd, b_kmajor=False)
        dx = torch.empty_like(dxd)
        lib.dropout_bwd(dxd.data_ptr(), dx.data_ptr(), B * Cin, p, seed, 1, s)
        if mod.grad_ready_hook:
            mod.grad_ready_hook(fl, 0, fl.total)
        return dx, None, None
