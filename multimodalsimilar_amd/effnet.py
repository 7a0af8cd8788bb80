"""MI355X-native EfficientNet image tower (the arithmetic of timm ``efficientnet_b0`` / ``efficientnet_b4`` that
cv_classifier.py:23-27,49 calls; architecture per SURVEY.md Appendix C).

Host-side schedule over the HIP kernels in csrc/conv.hip, csrc/mbconv.hip and csrc/gemm.hip.  Activations are NHWC so a 1x1
conv is a GEMM over [pixels, channels]; every BatchNorm is train-mode (batch statistics); the BN+SiLU(+SE gate)
that precedes the projection conv is applied while that GEMM stages its operand, so the widest tensors are
written once and never re-materialised activated.  Parameter names / shapes are timm's, so state dicts
interchange; all parameters live in one flat buffer (fp32 master, fp32 grad, bf16 + fp16 shadows).

Element types: every FORWARD tensor (conv outputs, activations, block outputs) and the weight copy the forward products read
are fp16 -- same bytes and MFMA rate as bf16, 11-bit significand: bf16 storage alone moved the embedding by 3.5-5 % against
north_star's 1e-2, fp16 moves it by ~0.6 % (oracle/effnet_ref.py emulate="fp16"); every GRADIENT tensor and the weight copy
the data-gradient products read are bf16 (per-element gradients of the 112x112 maps sit below fp16's normal range).
"""
import math
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .flat import FlatBuffer
from ._lib import MmsimError, lib

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

_BASE = [("ds", 1, 3, 1, 1, 16), ("ir", 2, 3, 2, 6, 24), ("ir", 2, 5, 2, 6, 40), ("ir", 3, 3, 2, 6, 80),
         ("ir", 3, 5, 1, 6, 112), ("ir", 4, 5, 2, 6, 192), ("ir", 1, 3, 1, 6, 320)]
_SCALE = {"efficientnet_b0": (1.0, 1.0), "efficientnet_b4": (1.4, 1.8)}


def _make_divisible(v, divisor=8, round_limit=0.9):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def build_arch(model_name):
    if model_name.startswith("tf_"):
        model_name = model_name[3:]     # the scripts only ever build the non-tf variant
    if model_name not in _SCALE:
        raise ValueError(f"unsupported image tower {model_name!r} (efficientnet_b0 / efficientnet_b4)")
    w, d = _SCALE[model_name]
    stem, head = _make_divisible(32 * w), _make_divisible(1280 * w)
    blocks, cin = [], stem
    for si, (typ, r, k, s, e, c) in enumerate(_BASE):
        cout = _make_divisible(c * w)
        for bi in range(int(math.ceil(r * d))):
            stride = s if bi == 0 else 1
            blocks.append(SimpleNamespace(type=typ, cin=cin, mid=cin * e, cout=cout, k=k, stride=stride,
                                          rd=int(round(cin * 0.25)), skip=(stride == 1 and cin == cout),
                                          name=f"blocks.{si}.{bi}"))
            cin = cout
    return SimpleNamespace(stem=stem, head=head, blocks=blocks, last=cin)


def count_macs(model_name, res=224):
    """Multiply-accumulates of one forward pass through the backbone (convolutions + SE MLPs), for bench.py's flop count."""
    a = build_arch(model_name)
    h = res // 2
    macs = h * h * a.stem * 27
    for b in a.blocks:
        ho = h // b.stride
        if b.type == "ir":
            macs += h * h * b.cin * b.mid
        macs += ho * ho * b.mid * b.k ** 2 + 2 * b.mid * b.rd + ho * ho * b.mid * b.cout
        h = ho
    return macs + h * h * a.last * a.head


# MMSIM_KEEP_A2=0: do not keep the activated depthwise output; the projection conv then applies BN + SiLU + gate while it
# stages its operand (saves one [pixels, mid] bf16 tensor per block, costs ~2 ms/step at cfg4: those products become VALU-bound)
_KEEP_A2 = os.environ.get("MMSIM_KEEP_A2", "1") != "0"
# MMSIM_A2_FLY=0: keep a2 also in the blocks whose projection conv runs the streaming kernels (A/B switch)
_A2_FLY = os.environ.get("MMSIM_A2_FLY", "1") != "0"
# MMSIM_DWTILE=0: the round-1 depthwise kernels (rows straight from global memory; separate bn_apply / bn_bwd_apply passes)
# instead of the LDS-tiled ones of csrc/mbconv.hip (A/B switch)
_DWTILE = os.environ.get("MMSIM_DWTILE", "1") != "0"
# MMSIM_DW5M=0: the 5 x 5 stride-1 depthwise blocks on the VALU tile kernels (mbconv.hip) instead of the matrix-core kernels (dwmfma.hip)
_DW5M = os.environ.get("MMSIM_DW5M", "1") != "0"
# MMSIM_DW5M_BWD: the fused backward on the matrix cores too -- 1 (default): where it measured faster than the VALU tile kernel (planes of
# at least 14 x 14: 379 vs 401 us at 28^2 x 336, 270 vs 335 us at 14^2 x 960; at 7^2 x 1632 it is 172 vs 151 us and stays off), 2: every
# eligible shape, 0: never
_DW5M_BWD = int(os.environ.get("MMSIM_DW5M_BWD", "1"))
# MMSIM_DWT_BWD_S2=0: the stride-2 blocks' depthwise backward as the round-1 kernels (bn_bwd_apply + dwconv_bwd_weight_xf + dwconv_bwd_data)
_DWT_BWD_S2 = os.environ.get("MMSIM_DWT_BWD_S2", "1") != "0"
# MMSIM_PW_FUSED=0: expand-stage backward as bn_bwd + two GEMMs instead of the one-pass mmsim_pw_expand_bwd (A/B switch)
_PW_FUSED = os.environ.get("MMSIM_PW_FUSED", "1") != "0"
# MMSIM_PW_PROJECT=0: projection conv of the early stages through the generic GEMM instead of the streaming kernels (A/B switch)
_PW_PROJECT = os.environ.get("MMSIM_PW_PROJECT", "1") != "0"
# MMSIM_S2_XF=0: stride-2 blocks store a1 = silu(bn1(z1)) (a bn_apply pass) instead of re-forming it in the depthwise kernels
_S2_XF = os.environ.get("MMSIM_S2_XF", "1") != "0"
# MMSIM_BN_FUSED_FINALIZE=0: BatchNorm statistics reduced with atomics and finalised by a second launch (A/B switch)
_BN_FUSED_FINALIZE = os.environ.get("MMSIM_BN_FUSED_FINALIZE", "1") != "0"

class _Holder(nn.Module):
    pass


def _node(root, dotted):
    mod = root
    for p in dotted.split("."):
        if p not in mod._modules:
            mod.add_module(p, _Holder())
        mod = mod._modules[p]
    return mod


def _bn_names(b):
    """(expand BN or None, depthwise BN, project BN) attribute names inside a block."""
    return (None, "bn1", "bn2") if b.type == "ds" else ("bn1", "bn2", "bn3")


class EfficientNet(nn.Module):
    """Backbone with timm's attribute names.  ``num_features`` = head width (1280 / 1792)."""

    def __init__(self, model_name="efficientnet_b4", seed=None, device=None):
        super().__init__()
        self.model_name = model_name
        a = self.arch = build_arch(model_name)
        self.num_features = a.head
        specs, bns = [], []

        def conv(name, shape):
            specs.append((name + ".weight", shape))

        def bn(name, c):
            specs.append((name + ".weight", (c,)))
            specs.append((name + ".bias", (c,)))
            bns.append((name, c))

        conv("conv_stem", (a.stem, 3, 3, 3)); bn("bn1", a.stem)
        for b in a.blocks:
            n = b.name
            e_bn, d_bn, p_bn = _bn_names(b)
            if b.type == "ir":
                conv(n + ".conv_pw", (b.mid, b.cin, 1, 1)); bn(n + "." + e_bn, b.mid)
            conv(n + ".conv_dw", (b.mid, 1, b.k, b.k)); bn(n + "." + d_bn, b.mid)
            specs.append((n + ".se.conv_reduce.weight", (b.rd, b.mid, 1, 1)))
            specs.append((n + ".se.conv_reduce.bias", (b.rd,)))
            specs.append((n + ".se.conv_expand.weight", (b.mid, b.rd, 1, 1)))
            specs.append((n + ".se.conv_expand.bias", (b.mid,)))
            conv(n + (".conv_pw" if b.type == "ds" else ".conv_pwl"), (b.cout, b.mid, 1, 1)); bn(n + "." + p_bn, b.cout)
        conv("conv_head", (a.head, a.last, 1, 1)); bn("bn2", a.head)
        self._bn_list = bns
        self._flat = FlatBuffer(specs, device="cpu", f16_shadow=True)
        self._init_weights(seed)
        for name, _ in specs:
            path, leaf = name.rsplit(".", 1)
            _node(self, path).register_parameter(leaf, nn.Parameter(self._flat.view(name)))
        # BatchNorm running statistics: views of one buffer each, so a step touches them with O(1) torch calls
        self._bn_off, off = {}, 0
        for n, c in bns:
            self._bn_off[n] = off
            off += c
        self._bn_total = off
        self._run = torch.cat([torch.zeros(off), torch.ones(off)])           # [mean | var]
        self._nbt = torch.zeros(len(bns), dtype=torch.long)
        for i, (n, c) in enumerate(bns):
            node = _node(self, n)
            o = self._bn_off[n]
            node.register_buffer("running_mean", self._run[o:o + c])
            node.register_buffer("running_var", self._run[off + o:off + o + c])
            node.register_buffer("num_batches_tracked", self._nbt[i])
        self._anchor = torch.zeros((), requires_grad=True)
        self._scratch = {}
        self._step_seed = 0
        self.grad_ready_hook = None
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------ parameters / buffers
    def _init_weights(self, seed):
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        for n in self._flat.names:
            v = self._flat.view(n)
            if v.dim() == 4:                      # timm efficientnet init: normal(0, sqrt(2 / fan_out))
                cout, cin_g, kh, kw = v.shape
                groups = cout if (cin_g == 1 and "conv_dw" in n) else 1
                fan_out = kh * kw * cout // groups
                v.copy_(torch.randn(v.shape, generator=g) * math.sqrt(2.0 / fan_out))
            elif n.endswith(".weight"):           # BN gamma
                v.fill_(1.0)
            else:
                v.zero_()

    def _rebind(self):
        named = dict(self.named_parameters())
        for n in self._flat.names:
            p = named[n]
            p.data = self._flat.view(n)
            p.grad = None
        off = self._bn_total
        for i, (n, c) in enumerate(self._bn_list):
            node = _node(self, n)
            o = self._bn_off[n]
            node._buffers["running_mean"] = self._run[o:o + c]
            node._buffers["running_var"] = self._run[off + o:off + o + c]
            node._buffers["num_batches_tracked"] = self._nbt[i]

    def _apply(self, fn, recurse=True):
        self._flat.apply_(fn)
        self._run = fn(self._run)
        self._nbt = fn(self._nbt)
        self._anchor = fn(self._anchor.detach()).requires_grad_(True)
        self._rebind()
        self._scratch = {}
        return self

    def _bind_grads(self):
        self._flat.ensure_device_state()
        named = dict(self.named_parameters())
        for n in self._flat.names:
            p = named[n]
            if p.grad is None or p.grad.data_ptr() != self._flat.gview(n).data_ptr():
                p.grad = self._flat.gview(n)

    def flat_buffers(self):
        return [self._flat]

    def sync_weights(self):
        self._flat.sync_shadow(force=True)

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._flat._shadow_version = None

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_scratch"] = {}
        st["grad_ready_hook"] = None
        return st

    def _buf(self, key, shape, dtype, zero=False):
        k = (key, tuple(shape), dtype)
        t = self._scratch.get(k)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self._flat.master.device)
            self._scratch[k] = t
        return t

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        """timm-style feature map [B, head, H/32, W/32] (fp32, NCHW).  CvClassifier uses forward_pooled instead."""
        raise NotImplementedError("use forward_pooled(): the un-pooled activated head feature map is never "
                                  "materialised on the HIP path (global pool is fused into the head BN+SiLU)")

    def forward_pooled(self, x):
        """AdaptiveAvgPool2d(1)(backbone(x)).view(B,-1) -> fp32 [B, num_features]  (cv_classifier.py:49-50)."""
        if not x.is_cuda:
            raise MmsimError("EfficientNet: inputs must be on the GPU; the HIP path has no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError("EfficientNet: expected an NCHW image batch [B,3,H,W] with H, W multiples of 32")
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if need_grad:
            return _EffFn.apply(self._anchor, self, x)
        st = self._run_forward(x)
        return st.pooled.clone()

    # ---- deferred launch: the two-tower model wants this tower's autograd node created BEFORE the text tower's (autograd
    # then replays the text backward first, whose few long GEMM launches let the host run ahead) but its forward kernels
    # enqueued AFTER the text tower's (same reason, forward direction).  Between defer_launches() and flush_deferred() a
    # training forward only creates the node and returns the (not yet written) pooled-feature tensor.
    _defer = None

    def defer_launches(self):
        self._defer = []

    def flush_deferred(self):
        if self._defer is None:
            raise MmsimError("EfficientNet.flush_deferred() without defer_launches()")
        pend, self._defer = self._defer, None
        if len(pend) > 1:
            raise MmsimError("EfficientNet: more than one deferred forward (their BatchNorm scratch would alias)")
        with torch.no_grad():          # `out` already carries the node: an in-place write under grad mode would rebase its history
            for x, out, holder in pend:
                holder.st = self._run_forward(x)
                out.copy_(holder.st.pooled)

    SCRATCH_FLOATS = 8 << 20      # per-block partial slabs of the channel reductions (largest need: ~5.5 M floats)

    def _scr(self):
        t = self._buf("scratch", (self.SCRATCH_FLOATS,), torch.float32)
        return t.data_ptr(), self.SCRATCH_FLOATS

    def _bnp(self, st, name, i):
        """per-BN fp32 scratch row i of (mean, rstd, scale, shift)."""
        o, c = self._bn_off[name], dict(self._bn_list)[name]
        return st.bnstat[i, o:o + c]

    def _bn_arm(self, st, name, sums, count):
        """Training mode: attach this BatchNorm's finalisation to the reduction of its statistics inside the producer that is about to
        run (mmsim_bn_finalize_arm); _bn_finalize after the producer then only flushes (a launch only if the producer did not reduce)."""
        if not self.training or not _BN_FUSED_FINALIZE:
            return
        fl = self._flat
        o, c = self._bn_off[name], sums.numel() // 2
        lib.bn_finalize_arm(sums.data_ptr(), fl.view(name + ".weight").data_ptr(), fl.view(name + ".bias").data_ptr(),
                            self._bnp(st, name, 0).data_ptr(), self._bnp(st, name, 1).data_ptr(),
                            self._bnp(st, name, 2).data_ptr(), self._bnp(st, name, 3).data_ptr(),
                            self._run[o:o + c].data_ptr(), self._run[self._bn_total + o:self._bn_total + o + c].data_ptr(),
                            c, float(count), BN_EPS, BN_MOMENTUM)

    def _bn_finalize(self, st, name, sums, count):
        fl, s = self._flat, ops._stream()
        o, c = self._bn_off[name], sums.numel() // 2
        if self.training and _BN_FUSED_FINALIZE:
            lib.bn_finalize_flush(s)
            return
        if not self.training:
            # eval: normalise with the running statistics (nn.BatchNorm2d.eval()); per-channel vectors only
            rm, rv = self._run[o:o + c], self._run[self._bn_total + o:self._bn_total + o + c]
            rstd = torch.rsqrt(rv + BN_EPS)
            scale = fl.view(name + ".weight") * rstd
            self._bnp(st, name, 0).copy_(rm); self._bnp(st, name, 1).copy_(rstd)
            self._bnp(st, name, 2).copy_(scale); self._bnp(st, name, 3).copy_(fl.view(name + ".bias") - rm * scale)
            return
        lib.bn_finalize(sums.data_ptr(), fl.view(name + ".weight").data_ptr(), fl.view(name + ".bias").data_ptr(),
                        self._bnp(st, name, 0).data_ptr(), self._bnp(st, name, 1).data_ptr(),
                        self._bnp(st, name, 2).data_ptr(), self._bnp(st, name, 3).data_ptr(),
                        self._run[o:o + c].data_ptr(), self._run[self._bn_total + o:self._bn_total + o + c].data_ptr(),
                        c, float(count), BN_EPS, BN_MOMENTUM, s)

    def _sums(self, st, name, which):
        o, c = self._bn_off[name], dict(self._bn_list)[name]
        buf = st.sums_f if which == "f" else st.sums_b
        return buf[2 * o:2 * o + 2 * c]

    def _block_fwd(self, st, b, cur, H, W):
        """One MBConv block: (expand 1x1 -> BN -> SiLU) -> depthwise -> BN -> SiLU -> SE -> project 1x1 -> BN (+skip)."""
        fl, s = self._flat, ops._stream()
        B, dev, bf = st.B, cur.device, torch.float16          # forward tensors: fp16 (module docstring)
        E = lambda *sh, dt=bf: torch.empty(*sh, dtype=dt, device=dev)
        V, SV = fl.view, fl.sview16                               # forward products read the fp16 weight shadow
        n = b.name
        e_bn, d_bn, p_bn = _bn_names(b)
        bs = SimpleNamespace(x_in=cur, H=H, W=W)
        P_in = B * H * W
        if b.type == "ir":
            bs.z1 = E(P_in, b.mid)
            sm = self._sums(st, n + "." + e_bn, "f")
            self._bn_arm(st, n + "." + e_bn, sm, P_in)
            w1 = SV(n + ".conv_pw.weight", (b.mid, b.cin))
            if _PW_PROJECT and lib.pw_expand_fwd_eligible(P_in, b.mid, b.cin):
                lib.pw_expand_fwd(cur.data_ptr(), w1.data_ptr(), bs.z1.data_ptr(), sm.data_ptr(), P_in, b.mid, b.cin, *self._scr(), s)
            else:
                lib.gemm_bf16_bnstats(0, P_in, b.mid, b.cin, cur.data_ptr(), b.cin, w1.data_ptr(), b.cin, bs.z1.data_ptr(), b.mid,
                                      None, None, None, 1, sm.data_ptr(), *self._scr(), s)     # conv + the BN statistics of z1
            self._bn_finalize(st, n + "." + e_bn, sm, P_in)
            if _DWTILE and (b.stride == 1 or _S2_XF):
                bs.a1 = None          # a1 = silu(bn(z1)) is formed inside the depthwise kernels, never materialised
            else:
                bs.a1 = E(P_in, b.mid)
                lib.bn_apply(bs.z1.data_ptr(), self._bnp(st, n + "." + e_bn, 2).data_ptr(),
                             self._bnp(st, n + "." + e_bn, 3).data_ptr(), None, bs.a1.data_ptr(), P_in, b.mid, 1, s)
        else:
            bs.a1 = cur
        Ho, Wo = (H + b.stride - 1) // b.stride, (W + b.stride - 1) // b.stride
        P_out = B * Ho * Wo
        if not hasattr(st, "wT_all"):             # a block driven on its own (tests): _run_forward makes this copy for all blocks at once
            self._make_wT(st, cur.device, s)
        o = self._tap_off[n]                      # tap-major copy of the depthwise weights
        bs.wT = st.wT_all[o:o + b.k * b.k * b.mid].view(b.k * b.k, b.mid)
        bs.z2 = E(P_out, b.mid)
        sm = self._sums(st, n + "." + d_bn, "f")
        self._bn_arm(st, n + "." + d_bn, sm, P_out)
        if not _DWTILE:
            lib.dwconv_fwd(bs.a1.data_ptr(), bs.wT.data_ptr(), bs.z2.data_ptr(), sm.data_ptr(), B, H, W, b.mid, b.k, b.stride, *self._scr(), s)
        elif bs.a1 is None and _DW5M and lib.dw5m_eligible(B, H, W, b.mid, b.k, b.stride):
            # 5 x 5 stride-1 blocks: the depthwise conv on the matrix cores (csrc/dwmfma.hip)
            lib.dw5m_fwd(bs.z1.data_ptr(), self._bnp(st, n + "." + e_bn, 2).data_ptr(), self._bnp(st, n + "." + e_bn, 3).data_ptr(),
                         bs.wT.data_ptr(), bs.z2.data_ptr(), sm.data_ptr(), B, H, W, b.mid, *self._scr(), s)
        elif bs.a1 is None:
            lib.dwtile_fwd(bs.z1.data_ptr(), self._bnp(st, n + "." + e_bn, 2).data_ptr(), self._bnp(st, n + "." + e_bn, 3).data_ptr(),
                           bs.wT.data_ptr(), bs.z2.data_ptr(), sm.data_ptr(), B, H, W, b.mid, b.k, b.stride, *self._scr(), s)
        else:
            lib.dwtile_fwd(bs.a1.data_ptr(), None, None, bs.wT.data_ptr(), bs.z2.data_ptr(), sm.data_ptr(), B, H, W, b.mid, b.k,
                           b.stride, *self._scr(), s)
        self._bn_finalize(st, n + "." + d_bn, sm, P_out)
        sc2, sh2 = self._bnp(st, n + "." + d_bn, 2), self._bnp(st, n + "." + d_bn, 3)
        bs.s = E(B, b.mid, dt=torch.float32)
        pw = n + (".conv_pw" if b.type == "ds" else ".conv_pwl")
        # early stages: both projection kernels stream and can form silu(bn(z2)) * gate themselves -- a2 is then not stored at all
        bs.a2_fly = (_A2_FLY and _PW_PROJECT and lib.pw_project_fwd_eligible(P_out, Ho * Wo, b.mid, b.cout)
                     and lib.pw_project_bwd_eligible(P_out, Ho * Wo, b.mid, b.cout))
        if _KEEP_A2 and not bs.a2_fly:      # the SE squeeze also keeps a2 = silu(bn(z2)): the projection conv's operand is then a2 * gate
            bs.a2 = E(P_out, b.mid)
            lib.pool_bn_act_store(bs.z2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), bs.a2.data_ptr(), bs.s.data_ptr(), B, Ho * Wo,
                                  b.mid, 1.0 / (Ho * Wo), s)
        else:
            bs.a2 = None
            lib.pool_bn_act(bs.z2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), None, bs.s.data_ptr(), B, Ho * Wo, b.mid, 1,
                            1.0 / (Ho * Wo), s)
        bs.hr = E(B, b.rd, dt=torch.float32)
        bs.hs = E(B, b.rd, dt=torch.float32)
        bs.gate = E(B, b.mid, dt=torch.float32)
        o = self._se_off[n]                       # conv_expand.weight transposed: made with the tap-major copies (_make_wT)
        bs.weT = st.wT_all[o:o + b.rd * b.mid].view(b.rd, b.mid)
        lib.se_mlp_fwd(bs.s.data_ptr(), V(n + ".se.conv_reduce.weight").data_ptr(), V(n + ".se.conv_reduce.bias").data_ptr(),
                       None, V(n + ".se.conv_expand.bias").data_ptr(),
                       bs.weT.data_ptr(), bs.hr.data_ptr(), bs.hs.data_ptr(), bs.gate.data_ptr(), B, b.mid, b.rd, s)
        pw = n + (".conv_pw" if b.type == "ds" else ".conv_pwl")
        bs.z3 = E(P_out, b.cout)
        w3 = SV(pw + ".weight", (b.cout, b.mid))
        sm = self._sums(st, n + "." + p_bn, "f")
        self._bn_arm(st, n + "." + p_bn, sm, P_out)
        if bs.a2_fly:
            lib.pw_project_fwd_xf(bs.z2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), bs.gate.data_ptr(), w3.data_ptr(), bs.z3.data_ptr(),
                                  sm.data_ptr(), P_out, Ho * Wo, b.mid, b.cout, *self._scr(), s)
        elif bs.a2 is not None and _PW_PROJECT and lib.pw_project_fwd_eligible(P_out, Ho * Wo, b.mid, b.cout):
            # early stages: one streaming pass over a2 (gate applied on the way into LDS), W3 resident in LDS, statistics in registers
            lib.pw_project_fwd(bs.a2.data_ptr(), bs.gate.data_ptr(), w3.data_ptr(), bs.z3.data_ptr(), sm.data_ptr(), P_out, Ho * Wo,
                               b.mid, b.cout, *self._scr(), s)
        elif bs.a2 is not None:
            lib.gemm_bf16_bnstats(1, P_out, b.cout, b.mid, bs.a2.data_ptr(), b.mid, w3.data_ptr(), b.mid, bs.z3.data_ptr(), b.cout,
                                  None, None, bs.gate.data_ptr(), Ho * Wo, sm.data_ptr(), *self._scr(), s)
        else:
            lib.gemm_bf16_bnstats(1, P_out, b.cout, b.mid, bs.z2.data_ptr(), b.mid, w3.data_ptr(), b.mid, bs.z3.data_ptr(), b.cout,
                                  sc2.data_ptr(), sh2.data_ptr(), bs.gate.data_ptr(), Ho * Wo, sm.data_ptr(), *self._scr(), s)
        self._bn_finalize(st, n + "." + p_bn, sm, P_out)
        nxt = E(P_out, b.cout)
        lib.bn_apply(bs.z3.data_ptr(), self._bnp(st, n + "." + p_bn, 2).data_ptr(), self._bnp(st, n + "." + p_bn, 3).data_ptr(),
                     cur.data_ptr() if b.skip else None, nxt.data_ptr(), P_out, b.cout, 0, s)
        bs.Ho, bs.Wo = Ho, Wo
        st.blocks.append(bs)
        return nxt, Ho, Wo

    def _run_forward(self, x):
        a, fl = self.arch, self._flat
        fl.sync_shadow()
        s = ops._stream()
        x = x.contiguous().float()
        B, _, Hi, Wi = x.shape
        dev = x.device
        bf = torch.float16                                        # forward tensors: fp16
        self._step_seed += 1
        self._gen = getattr(self, "_gen", 0) + 1      # BN statistics live in module scratch (bnstat): see _run_backward
        st = SimpleNamespace(B=B, Hi=Hi, Wi=Wi, x=x, blocks=[], gen=self._gen)
        st.bnstat = self._buf("bnstat", (4, self._bn_total), torch.float32)
        self._make_wT(st, dev, s)
        st.sums_f = self._buf("sums_f", (2 * self._bn_total,), torch.float32)
        st.sums_f.zero_()
        E = lambda *sh, dt=bf: torch.empty(*sh, dtype=dt, device=dev)
        V, SV = fl.view, fl.sview16
        # ---- stem
        H, W = Hi // 2, Wi // 2
        P = B * H * W
        st.z0 = E(P, a.stem)
        self._bn_arm(st, "bn1", self._sums(st, "bn1", "f"), P)
        lib.stem_fwd(x.data_ptr(), V("conv_stem.weight").data_ptr(), st.z0.data_ptr(), self._sums(st, "bn1", "f").data_ptr(),
                     B, Hi, Wi, a.stem, *self._scr(), s)
        self._bn_finalize(st, "bn1", self._sums(st, "bn1", "f"), P)
        cur = E(P, a.stem)
        lib.bn_apply(st.z0.data_ptr(), self._bnp(st, "bn1", 2).data_ptr(), self._bnp(st, "bn1", 3).data_ptr(), None,
                     cur.data_ptr(), P, a.stem, 1, s)
        st.x0 = cur
        # ---- MBConv blocks
        for b in a.blocks:
            cur, H, W = self._block_fwd(st, b, cur, H, W)
        # ---- head conv + BN + SiLU + global average pool (fused into the pooling kernel)
        P = B * H * W
        st.x_last, st.Hh, st.Wh = cur, H, W
        st.zh = E(P, a.head)
        sm = self._sums(st, "bn2", "f")
        self._bn_arm(st, "bn2", sm, P)
        wh = SV("conv_head.weight", (a.head, a.last))
        lib.gemm_bf16_bnstats(0, P, a.head, a.last, cur.data_ptr(), a.last, wh.data_ptr(), a.last, st.zh.data_ptr(), a.head,
                              None, None, None, 1, sm.data_ptr(), *self._scr(), s)
        self._bn_finalize(st, "bn2", sm, P)
        st.pooled = E(B, a.head, dt=torch.float32)
        lib.pool_bn_act(st.zh.data_ptr(), self._bnp(st, "bn2", 2).data_ptr(), self._bnp(st, "bn2", 3).data_ptr(), None,
                        st.pooled.data_ptr(), B, H * W, a.head, 1, 1.0 / (H * W), s)
        if self.training:
            self._nbt += 1
        return st

    # ------------------------------------------------------------------ backward
    def _tap_tables(self, dev):
        """Segment tables of mmsim_dw_tap_major_batch: every block's conv_dw.weight <-> its slice of the contiguous tap-major buffers."""
        if getattr(self, "_tap_dev", None) == dev:
            return
        fl = self._flat
        self._tap_off, self._se_off, fwd, bwd, tot, mx = {}, {}, [], [], 0, 0
        for blk in self.arch.blocks:
            kk, po = blk.k * blk.k, fl.offsets[blk.name + ".conv_dw.weight"]
            self._tap_off[blk.name] = tot
            fwd.append([po, tot, blk.mid, kk])          # master weights -> wT_all
            bwd.append([tot, po, blk.mid, kk])          # gT_all -> gradient buffer
            tot += kk * blk.mid
            mx = max(mx, kk * blk.mid)
        for blk in self.arch.blocks:                    # the SE expand weights [mid][rd] <-> [rd][mid] ride in the same two buffers
            po = fl.offsets[blk.name + ".se.conv_expand.weight"]
            self._se_off[blk.name] = tot
            fwd.append([po, tot, blk.mid, blk.rd])
            bwd.append([tot, po, blk.mid, blk.rd])
            tot += blk.rd * blk.mid
            mx = max(mx, blk.rd * blk.mid)
        self._tap_total, self._tap_max, self._tap_n = tot, mx, len(fwd)
        self._tap_fwd = torch.tensor(fwd, dtype=torch.int64, device=dev)
        self._tap_bwd = torch.tensor(bwd, dtype=torch.int64, device=dev)
        self._tap_dev = dev

    def _make_wT(self, st, dev, s):
        """Tap-major fp32 copies of every block's depthwise weights: one launch per forward."""
        self._tap_tables(dev)
        st.wT_all = self._buf("wT_all", (self._tap_total,), torch.float32)
        lib.dw_tap_major_batch(self._tap_fwd.data_ptr(), self._tap_n, self._flat.master.data_ptr(), st.wT_all.data_ptr(), 1,
                               self._tap_max, s)

    def _zero_gT(self, st):
        """Tap-major depthwise weight gradients and transposed SE expand-weight gradients of all blocks: one zeroed buffer per step
        (same slice layout as wT_all, _tap_tables) instead of a fill per block."""
        self._tap_tables(self._flat.master.device)
        st.gT_off = self._tap_off
        st.gT_all = self._buf("gT_all", (self._tap_total,), torch.float32)
        st.gT_all.zero_()

    def _bn_bwd(self, st, name, dy, z, P, C, dz, act, gate=None, dsq=None, hw=1, sums_ready=False):
        fl = self._flat
        sums = self._sums(st, name, "b")
        lib.bn_bwd(dy.data_ptr(), z.data_ptr(), self._bnp(st, name, 0).data_ptr(), self._bnp(st, name, 1).data_ptr(),
                   self._bnp(st, name, 2).data_ptr(), self._bnp(st, name, 3).data_ptr(),
                   None if gate is None else gate.data_ptr(), None if dsq is None else dsq.data_ptr(), hw, int(act),
                   sums.data_ptr(), int(sums_ready), dz.data_ptr(), fl.gview(name + ".weight").data_ptr(),
                   fl.gview(name + ".bias").data_ptr(), P, C, *self._scr(), ops._stream())

    def _block_bwd(self, st, b, bs, dx):
        """Backward of one MBConv block: dx = dLoss/d(block output) -> returns dLoss/d(block input); parameter
        gradients are accumulated into the flat gradient buffer."""
        fl, s = self._flat, ops._stream()
        B, dev, bf = st.B, dx.device, torch.bfloat16
        E = lambda *sh, dt=bf: torch.empty(*sh, dtype=dt, device=dev)
        V, G, SV = fl.view, fl.gview, fl.sview
        n = b.name
        e_bn, d_bn, p_bn = _bn_names(b)
        pw = n + (".conv_pw" if b.type == "ds" else ".conv_pwl")
        Ho, Wo, Hn, Wn = bs.Ho, bs.Wo, bs.H, bs.W
        P_out, P_in = B * Ho * Wo, B * Hn * Wn
        dz3 = E(P_out, b.cout)
        self._bn_bwd(st, n + "." + p_bn, dx, bs.z3, P_out, b.cout, dz3, act=False)
        sc2, sh2 = self._bnp(st, n + "." + d_bn, 2), self._bnp(st, n + "." + d_bn, 3)
        gw3 = G(pw + ".weight").view(b.cout, b.mid)
        da2g = E(P_out, b.mid)
        if getattr(bs, "a2_fly", False):
            lib.pw_project_bwd_xf(dz3.data_ptr(), bs.z2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), bs.gate.data_ptr(),
                                  SV(pw + ".weight", (b.cout, b.mid)).data_ptr(), da2g.data_ptr(), gw3.data_ptr(), P_out, Ho * Wo,
                                  b.mid, b.cout, *self._scr(), s)
        elif getattr(bs, "a2", None) is not None and _PW_PROJECT and lib.pw_project_bwd_eligible(P_out, Ho * Wo, b.mid, b.cout):
            # early stages: dW3 and d(a2*gate) out of one streaming pass over a2
            lib.pw_project_bwd(dz3.data_ptr(), bs.a2.data_ptr(), bs.gate.data_ptr(), SV(pw + ".weight", (b.cout, b.mid)).data_ptr(),
                               da2g.data_ptr(), gw3.data_ptr(), P_out, Ho * Wo, b.mid, b.cout, *self._scr(), s)
        else:
            with ops.gemm_group():       # dW3 and d(a2*gate) share one launch
                if getattr(bs, "a2", None) is not None:
                    lib.gemm_bf16_xf(2, b.cout, b.mid, P_out, dz3.data_ptr(), b.cout, bs.a2.data_ptr(), b.mid, gw3.data_ptr(), b.mid, 1,
                                     None, None, bs.gate.data_ptr(), Ho * Wo, ops.pick_split_k(b.cout, b.mid, P_out), 1, s)
                else:
                    lib.gemm_bf16_xf(2, b.cout, b.mid, P_out, dz3.data_ptr(), b.cout, bs.z2.data_ptr(), b.mid, gw3.data_ptr(), b.mid, 1,
                                     sc2.data_ptr(), sh2.data_ptr(), bs.gate.data_ptr(), Ho * Wo,
                                     ops.pick_split_k(b.cout, b.mid, P_out), 1, s)
                ops.gemm(dz3, SV(pw + ".weight", (b.cout, b.mid)), da2g, b_kmajor=False)
        # one pass over (z2, da2g): dgate for the SE backward + the partial sums of the depthwise BN's backward statistics
        out5 = E(5, B, b.mid, dt=torch.float32)
        mu2, rs2 = self._bnp(st, n + "." + d_bn, 0), self._bnp(st, n + "." + d_bn, 1)
        lib.pool_bn_bwd(bs.z2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), mu2.data_ptr(), rs2.data_ptr(), da2g.data_ptr(),
                        out5.data_ptr(), B, Ho * Wo, b.mid, s)
        dgate = out5[0]
        dr, ds = E(B, b.rd, dt=torch.float32), E(B, b.mid, dt=torch.float32)
        if not hasattr(st, "gT_all"):
            self._zero_gT(st)
        batched = getattr(st, "tap_batched", False)
        if batched:       # the transposed expand-weight gradient lands in its slice of the zeroed gT_all; transposed back with all the others
            o = self._se_off[n]
            dweT = st.gT_all[o:o + b.rd * b.mid]
        else:             # data parallel (the block's gradient range must be final when reported) or a lone block
            dweT = E(b.rd, b.mid, dt=torch.float32)
        lib.se_mlp_bwd(dgate.data_ptr(), bs.gate.data_ptr(), bs.hr.data_ptr(), bs.hs.data_ptr(), bs.s.data_ptr(),
                       V(n + ".se.conv_reduce.weight").data_ptr(), bs.weT.data_ptr(),
                       dr.data_ptr(), ds.data_ptr(), dweT.data_ptr(), G(n + ".se.conv_reduce.weight").data_ptr(),
                       G(n + ".se.conv_reduce.bias").data_ptr(), None if batched else G(n + ".se.conv_expand.weight").data_ptr(),
                       G(n + ".se.conv_expand.bias").data_ptr(), B, b.mid, b.rd, s)
        lib.bn_bwd_sums_from_pool(out5.data_ptr(), bs.gate.data_ptr(), ds.data_ptr(), self._sums(st, n + "." + d_bn, "b").data_ptr(),
                                  B, Ho * Wo, b.mid, s)
        if not hasattr(st, "gT_all"):
            self._zero_gT(st)
        gT = st.gT_all[st.gT_off[n]:st.gT_off[n] + b.k * b.k * b.mid]
        fused = _DWTILE and (b.stride == 1 or (_DWT_BWD_S2 and b.type == "ir"))
        if fused and b.stride == 2:
            # the stride-2 form of the same kernel (round 4): the centre tile in input space, dz2 staged on the output plane
            bnp = lambda nm, i: self._bnp(st, nm, i).data_ptr()
            dn, en = n + "." + d_bn, n + "." + e_bn
            dpre1 = E(P_in, b.mid)
            lib.dwtile_bwd_s2(da2g.data_ptr(), bs.z2.data_ptr(), bnp(dn, 2), bnp(dn, 3), bnp(dn, 0), bnp(dn, 1),
                              self._sums(st, dn, "b").data_ptr(), bs.gate.data_ptr(), ds.data_ptr(),
                              bs.z1.data_ptr(), bnp(en, 2), bnp(en, 3), bnp(en, 0), bnp(en, 1), bs.wT.data_ptr(),
                              dpre1.data_ptr(), self._sums(st, en, "b").data_ptr(), gT.data_ptr(), G(dn + ".weight").data_ptr(),
                              G(dn + ".bias").data_ptr(), B, Hn, Wn, b.mid, b.k, *self._scr(), s)
            del da2g
        elif fused:
            # ONE kernel: depthwise-BN + SiLU + gate backward (on the way into LDS), depthwise data and weight gradients,
            # expand-BN + SiLU backward on the way out (mmsim_dwtile_bwd)
            bnp = lambda nm, i: self._bnp(st, nm, i).data_ptr()
            dn = n + "." + d_bn
            common = (da2g.data_ptr(), bs.z2.data_ptr(), bnp(dn, 2), bnp(dn, 3), bnp(dn, 0), bnp(dn, 1),
                      self._sums(st, dn, "b").data_ptr(), bs.gate.data_ptr(), ds.data_ptr())
            if b.type == "ir":
                en = n + "." + e_bn
                dpre1 = E(P_in, b.mid)
                if _DW5M_BWD and (_DW5M_BWD == 2 or Hn * Wn >= 196) and lib.dw5m_eligible(B, Hn, Wn, b.mid, b.k, 1):       # 5 x 5 blocks: on the matrix cores (csrc/dwmfma.hip)
                    lib.dw5m_bwd(*common, bs.z1.data_ptr(), bnp(en, 2), bnp(en, 3), bnp(en, 0), bnp(en, 1), bs.wT.data_ptr(),
                                 dpre1.data_ptr(), self._sums(st, en, "b").data_ptr(), gT.data_ptr(), G(dn + ".weight").data_ptr(),
                                 G(dn + ".bias").data_ptr(), B, Hn, Wn, b.mid, *self._scr(), s)
                else:
                    lib.dwtile_bwd(*common, bs.z1.data_ptr(), bnp(en, 2), bnp(en, 3), bnp(en, 0), bnp(en, 1), None, bs.wT.data_ptr(),
                                   dpre1.data_ptr(), self._sums(st, en, "b").data_ptr(), gT.data_ptr(), G(dn + ".weight").data_ptr(),
                                   G(dn + ".bias").data_ptr(), B, Hn, Wn, b.mid, b.k, *self._scr(), s)
            else:
                dx_in = E(P_in, b.cin)
                lib.dwtile_bwd(*common, bs.x_in.data_ptr(), None, None, None, None, dx.data_ptr() if b.skip else None,
                               bs.wT.data_ptr(), dx_in.data_ptr(), None, gT.data_ptr(), G(dn + ".weight").data_ptr(),
                               G(dn + ".bias").data_ptr(), B, Hn, Wn, b.mid, b.k, *self._scr(), s)
            del da2g
        else:
            dz2 = E(P_out, b.mid)
            self._bn_bwd(st, n + "." + d_bn, da2g, bs.z2, P_out, b.mid, dz2, act=True, gate=bs.gate, dsq=ds, hw=Ho * Wo,
                         sums_ready=True)
            del da2g
            if bs.a1 is None:      # stride-2 block without a stored a1: the weight gradient re-forms it from z1
                en1 = n + "." + e_bn
                lib.dwconv_bwd_weight_xf(dz2.data_ptr(), bs.z1.data_ptr(), self._bnp(st, en1, 2).data_ptr(), self._bnp(st, en1, 3).data_ptr(),
                                         gT.data_ptr(), B, Hn, Wn, b.mid, b.k, b.stride, *self._scr(), s)
            else:
                lib.dwconv_bwd_weight(dz2.data_ptr(), bs.a1.data_ptr(), gT.data_ptr(), B, Hn, Wn, b.mid, b.k, b.stride, *self._scr(), s)
        if not getattr(st, "tap_batched", False):      # data parallel (this block's gradient range is reported below and must be
            lib.dw_grad_from_tap_major(gT.data_ptr(), G(n + ".conv_dw.weight").data_ptr(), b.mid, b.k, s)      # final), or a lone block
        if b.type == "ir":
            en = n + "." + e_bn
            if not fused:
                dpre1 = E(P_in, b.mid)
                lib.dwconv_bwd_data(dz2.data_ptr(), bs.wT.data_ptr(), bs.z1.data_ptr(), self._bnp(st, en, 0).data_ptr(),
                                    self._bnp(st, en, 1).data_ptr(), self._bnp(st, en, 2).data_ptr(), self._bnp(st, en, 3).data_ptr(),
                                    None, dpre1.data_ptr(), self._sums(st, en, "b").data_ptr(), B, Hn, Wn, b.mid, b.k, b.stride,
                                    *self._scr(), s)
            if _PW_FUSED and lib.pw_expand_bwd_eligible(P_in, b.mid, b.cin):
                # one streaming pass: bn1 backward applied on the way into LDS, dx and dW1 out of the same staged strip
                dx_in = E(P_in, b.cin)
                lib.pw_expand_bwd(dpre1.data_ptr(), bs.z1.data_ptr(), bs.x_in.data_ptr(), dx.data_ptr() if b.skip else None,
                                  SV(n + ".conv_pw.weight", (b.mid, b.cin)).data_ptr(), self._bnp(st, en, 2).data_ptr(),
                                  self._bnp(st, en, 0).data_ptr(), self._bnp(st, en, 1).data_ptr(),
                                  self._sums(st, en, "b").data_ptr(), dx_in.data_ptr(), G(n + ".conv_pw.weight").data_ptr(),
                                  G(en + ".weight").data_ptr(), G(en + ".bias").data_ptr(), P_in, b.mid, b.cin, *self._scr(), s)
                del dpre1
            else:
                dz1 = E(P_in, b.mid)
                self._bn_bwd(st, en, dpre1, bs.z1, P_in, b.mid, dz1, act=False, sums_ready=True)
                del dpre1
                dx_in = E(P_in, b.cin)
                with ops.gemm_group():       # dW1 and dx share one launch
                    ops.gemm(dz1, bs.x_in, G(n + ".conv_pw.weight").view(b.mid, b.cin), trans_a=True, b_kmajor=False,
                             split_k=ops.pick_split_k(b.mid, b.cin, P_in), accumulate=True)
                    ops.gemm(dz1, SV(n + ".conv_pw.weight", (b.mid, b.cin)), dx_in, b_kmajor=False,
                             epilogue=ops.EPI_ADD if b.skip else ops.EPI_NONE, aux_in=dx if b.skip else None)
        elif not fused:
            dx_in = E(P_in, b.cin)
            lib.dwconv_bwd_data(dz2.data_ptr(), bs.wT.data_ptr(), None, None, None, None, None,
                                dx.data_ptr() if b.skip else None, dx_in.data_ptr(), None, B, Hn, Wn, b.mid, b.k, b.stride,
                                    *self._scr(), s)
        if self.grad_ready_hook:
            first = n + (".conv_pw.weight" if b.type == "ir" else ".conv_dw.weight")
            self.grad_ready_hook(fl, *fl.span(first, n + "." + p_bn + ".bias"))
        return dx_in

    def _run_backward(self, st, dpooled):
        a, fl = self.arch, self._flat
        if st.gen != getattr(self, "_gen", 0):
            raise MmsimError("EfficientNet: backward of a forward whose BatchNorm statistics scratch was overwritten by a later "
                             "forward; run backward before the next forward of this tower")
        self._bind_grads()
        s = ops._stream()
        B = st.B
        dev = dpooled.device
        bf = torch.bfloat16
        E = lambda *sh, dt=bf: torch.empty(*sh, dtype=dt, device=dev)
        V, G, SV = fl.view, fl.gview, fl.sview
        st.sums_b = self._buf("sums_b", (2 * self._bn_total,), torch.float32)
        st.sums_b.zero_()
        self._zero_gT(st)
        dpooled = dpooled.contiguous().float()
        # ---- head
        H, W = st.Hh, st.Wh
        P = B * H * W
        dyh = E(P, a.head)
        lib.broadcast_pool_grad(dpooled.data_ptr(), dyh.data_ptr(), B, H * W, a.head, s)
        dzh = E(P, a.head)
        self._bn_bwd(st, "bn2", dyh, st.zh, P, a.head, dzh, act=True)
        dx = E(P, a.last)
        with ops.gemm_group():
            ops.gemm(dzh, st.x_last, G("conv_head.weight").view(a.head, a.last), trans_a=True, b_kmajor=False,
                     split_k=ops.pick_split_k(a.head, a.last, P), accumulate=True)
            ops.gemm(dzh, SV("conv_head.weight", (a.head, a.last)), dx, b_kmajor=False)
        del dyh, dzh
        # ---- blocks, last to first
        st.tap_batched = not self.grad_ready_hook
        for b, bs in zip(reversed(a.blocks), reversed(st.blocks)):
            dx = self._block_bwd(st, b, bs, dx)
        if st.tap_batched:      # every block's tap-major depthwise weight gradient into the gradient buffer, one launch
            lib.dw_tap_major_batch(self._tap_bwd.data_ptr(), self._tap_n, st.gT_all.data_ptr(), fl.grad.data_ptr(), 0, self._tap_max, s)
        # ---- stem
        P0 = B * (st.Hi // 2) * (st.Wi // 2)
        dz0 = E(P0, a.stem)
        self._bn_bwd(st, "bn1", dx, st.z0, P0, a.stem, dz0, act=True)
        lib.stem_wgrad(dz0.data_ptr(), st.x.data_ptr(), G("conv_stem.weight").data_ptr(), B, st.Hi, st.Wi, a.stem, *self._scr(), s)
        if self.grad_ready_hook:
            self.grad_ready_hook(fl, *fl.span("conv_stem.weight", "bn1.bias"))
            self.grad_ready_hook(fl, *fl.span("conv_head.weight", "bn2.bias"))


class _EffFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, x):
        holder = SimpleNamespace(st=None)
        ctx.model, ctx.holder = model, holder
        if model._defer is not None:
            # deferred launch (see EfficientNet.defer_launches): the autograd node exists now, the kernels are enqueued later
            out = torch.empty(x.shape[0], model.arch.head, dtype=torch.float32, device=x.device)
            model._defer.append((x, out, holder))
            return out
        holder.st = model._run_forward(x)
        return holder.st.pooled.clone()

    @staticmethod
    def backward(ctx, dpooled):
        if ctx.holder.st is None:
            raise MmsimError("EfficientNet: backward before flush_deferred() launched the deferred forward")
        ctx.model._run_backward(ctx.holder.st, dpooled)
        ctx.holder.st = None
        return None, None, None
