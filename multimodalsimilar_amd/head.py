"""ArcFace margin head and the two-tower glue on the HIP kernels.

ArcMarginProduct mirrors /root/reference/arcface.py:17-67 (same constructor, attributes, forward /
forward_test / update_m) and adds ``forward_loss`` -- the fused path the training entry point uses:
normalise -> bf16 MFMA cosine GEMM whose EPILOGUE leaves per-segment online-softmax statistics of the scaled margin logits
(mmsim_arcface_fwd_fused: no logits / softmax / one-hot tensors, no separate pass over the cosines) -> one combine per row (log-sum-exp,
loss, argmax) + the mean; backward: ONE pass over the cosines writes dcos (the upstream gradient read from device memory) and the
row vectors of the weight gradient's normalisation backward (mmsim_arcface_dcos_rowfix), then the two backward GEMMs.  The weight matrix is the big operand
([classes, D], 1.1 GB fp32 at 100 000 x 2816), so nothing passes over it on its own: F.normalize(weight) comes out of the
AdamW launch (or is reused while the weights are static) and its backward is the dW product's epilogue.
"""
import math
import os

import torch
import torch.nn as nn

from . import ops
from .flat import FlatBuffer
from ._lib import MmsimError


# MMSIM_HEAD_FUSED_DW=0: weight gradient through an fp32 dW_hat buffer + a separate normalise-backward pass (A/B switch)
_FUSED_DW = os.environ.get("MMSIM_HEAD_FUSED_DW", "1") != "0"
# MMSIM_HEAD_FUSED_NORM=0: F.normalize(weight) as its own pass in every forward instead of out of the AdamW launch
_FUSED_NORM = os.environ.get("MMSIM_HEAD_FUSED_NORM", "1") != "0"


def _margin_consts(m):
    return math.cos(m), math.sin(m), math.cos(math.pi - m), math.sin(math.pi - m) * m


class ArcMarginProduct(nn.Module):
    def __init__(self, in_feature=128, out_feature=10575, s=64.0, m=0.40, easy_margin=False):
        super().__init__()
        if in_feature % 8:
            raise ValueError("ArcMarginProduct: in_feature must be a multiple of 8 for the bf16 MFMA kernels")
        self.in_feature = in_feature
        self.out_feature = out_feature
        self.s = s
        self.m = m
        self._flat = FlatBuffer([("weight", (out_feature, in_feature))], device="cpu")
        w = self._flat.view("weight")
        nn.init.xavier_uniform_(w)                                                   # arcface.py:25
        self.weight = nn.Parameter(w)
        self._flat.overwrite_capable = self._flat.total == w.numel()      # the dW product covers the whole gradient buffer
        self.easy_margin = easy_margin
        self.cos_m, self.sin_m, self.th, self.mm = _margin_consts(m)                 # arcface.py:28-33
        self._scratch = {}
        self._wh_key = None           # (weight version, address) the cached F.normalize(weight) belongs to
        self._gen = 0                 # bumped by every _cosines(): autograd contexts check their scratch is still theirs
        self.grad_ready_hook = None

    def update_m(self, delta):                                                       # arcface.py:35-42
        updated_m = self.m + delta
        if updated_m >= 1e-6 and updated_m <= 1.0:
            self.m = updated_m
            self.cos_m, self.sin_m, self.th, self.mm = _margin_consts(self.m)

    # ---- flat-buffer plumbing (same contract as the towers)
    def _apply(self, fn, recurse=True):
        self._flat.apply_(fn)
        self.weight.data = self._flat.view("weight")
        self.weight.grad = None
        self._flat.overwrite_capable = self._flat.total == self._flat.view("weight").numel()    # no padding outside `weight`
        self._scratch = {}
        self._wh_key = None
        return self

    def flat_buffers(self):
        return [self._flat]

    def _bind_grads(self):
        fl = self._flat
        if fl.grad is None or fl.grad.device != fl.master.device:
            fl.grad = torch.zeros_like(fl.master)
        g = fl.gview("weight")
        if self.weight.grad is None or self.weight.grad.data_ptr() != g.data_ptr():
            self.weight.grad = g

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_scratch"] = {}
        st["_wh_key"] = None
        st["grad_ready_hook"] = None
        return st

    # ---- F.normalize(self.weight) out of the optimiser launch (FusedAdamW): see mmsim_adamw_rows_l2norm
    def adamw_row_buffers(self):
        C, D = self.out_feature, self.in_feature
        if not _FUSED_NORM or not self.weight.is_cuda or D % 4 or D > 4096:
            return None
        return C, D, self._wh_buf()[:C], self._buf("inv_w", (C,), torch.float32)

    def mark_normalised(self):
        self._wh_key = (self.weight._version, self.weight.data_ptr())

    def invalidate_normalised(self):
        self._wh_key = None

    def _cpad(self):
        """Class dimension of the scratch buffers (w_hat rows, cos / dcos leading dimension).  Large heads round it up to a multiple of
        256 so that the cosine product [B, C] and dxh = dcos w_hat (reduction over C) are tile-aligned for the pipelined LDS-DMA GEMM
        (C = 100 000 -> 100 096: zero rows of w_hat give zero cosines, zero pad columns of dcos add nothing); both products stream
        the 563 MB w_hat and ran the ragged-shape kernel at 1.7-1.9 TB/s, exposed between the towers' forward and backward."""
        C = self.out_feature
        return ops.round_up(C, 256) if C >= 4096 else ops.round_up(C, 8)

    def _wh_buf(self):
        return self._buf("wh", (self._cpad(), self.in_feature), torch.bfloat16, zero=True)      # rows >= C stay zero

    def _buf(self, name, shape, dtype, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._scratch.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.weight.device)
            self._scratch[key] = t
        return t

    # ---- kernels
    def _cosines(self, x, product=True):
        """x [B,D] f32 -> (cos f32 [B, ldc] view [:, :C], saved state).  product=False: everything but the cosine product itself (the
        fused forward runs it with the statistics epilogue)."""
        if not x.is_cuda:
            raise MmsimError("ArcMarginProduct: inputs must be on the GPU; the HIP path has no CPU fallback")
        if x.dim() != 2 or x.shape[1] != self.in_feature:
            raise ValueError(f"ArcMarginProduct: expected x [B, {self.in_feature}], got {tuple(x.shape)}")
        B, D, C = x.shape[0], self.in_feature, self.out_feature
        x = x.contiguous().float()
        ldc = self._cpad()
        whp = self._wh_buf()                                                         # [ldc, D], rows >= C zero
        wh = whp[:C]
        inv_w = self._buf("inv_w", (C,), torch.float32)
        xh = self._buf("xh", (B, D), torch.bfloat16)
        inv_x = self._buf("inv_x", (B,), torch.float32)
        cos = self._buf("cos", (B, ldc), torch.float32, zero=True)
        self._gen = getattr(self, "_gen", 0) + 1
        key = (self.weight._version, self.weight.data_ptr())
        if self._wh_key != key:               # otherwise w_hat is current: left by the last AdamW launch, or the weights are static
            ops.l2norm_fwd(self.weight.detach(), None, wh, 0, inv_w)                 # F.normalize(self.weight)
            self._wh_key = key
        ops.l2norm_fwd(x, None, xh, 0, inv_x)                                        # F.normalize(x)
        if product:
            ops.gemm(xh, whp, cos)                                                   # F.linear  (arcface.py:47); pad columns = 0
        return cos, dict(x=x, xh=xh, wh=wh, whp=whp, inv_x=inv_x, inv_w=inv_w, B=B, ldc=ldc, gen=self._gen)

    def _check_gen(self, st):
        # xh / inv_x / cos / dcos are module scratch reused by the next call: a backward that runs after another forward
        # would read the newer call's values
        if st["gen"] != self._gen:
            raise MmsimError("ArcMarginProduct: backward of a forward whose scratch buffers were overwritten by a later call; "
                             "run backward before the next forward of this head")

    def _backward_from_dcos(self, st, dcos, cos=None, rowvec=None):
        """dcos bf16 [B, ldc] (pad zero) -> dx f32 [B,D]; accumulates into weight.grad.  With the cosines at hand the backward of
        F.normalize(self.weight) is folded into the dcos^T x_hat product's epilogue (no fp32 dW_hat round trip, no second pass
        over W): w_hat . dW_hat = sum_b dcos cos, so both row vectors of the correction are known before the product runs."""
        B, D, C, ldc = st["B"], self.in_feature, self.out_feature, st["ldc"]
        self._bind_grads()
        dxh = self._buf("dxh", (B, D), torch.float32)
        dx = torch.empty((B, D), dtype=torch.float32, device=dcos.device)
        # dxh = dcos @ Wh: 2 x 22 output tiles over a 100k-long reduction -> split-K (fp32 atomics into the zeroed buffer)
        sk = ops.pick_split_k(B, D, ldc)
        if sk > 1:
            dxh.zero_()
        ops.gemm(dcos, st["whp"], dxh, b_kmajor=False, split_k=sk, accumulate=sk > 1)      # over the padded class range (zeros)
        ops.l2norm_bwd(st["x"], st["inv_x"], dxh, 0, dx)
        # this product writes every element of the head's gradient buffer: after a lazy zero_grad it overwrites (no fill, no read)
        acc = not self._flat.take_zero_pending()
        if cos is not None and _FUSED_DW:
            if rowvec is None:            # (the fused loss path's dcos pass has already left the two row vectors)
                rowvec = self._buf("rowvec", (2, C), torch.float32)
                ops.lib.arcface_rowfix(dcos.data_ptr(), cos.data_ptr(), ldc, st["inv_w"].data_ptr(), rowvec.data_ptr(), B, C, ops._stream())
            ops.gemm(dcos[:, :C], st["xh"], self._flat.gview("weight").view(C, D), trans_a=True, b_kmajor=False, bias=rowvec,
                     epilogue=ops.EPI_ROWFIX, aux_in=st["wh"], accumulate=acc)
        else:
            dwh = self._buf("dwh", (C, D), torch.float32)
            ops.gemm(dcos[:, :C], st["xh"], dwh, trans_a=True, b_kmajor=False)           # dWh = dcos^T @ xh
            ops.l2norm_bwd(self.weight.detach(), st["inv_w"], dwh, 0, self._flat.gview("weight"), accumulate=acc)
        if self.grad_ready_hook:
            self.grad_ready_hook(self._flat, 0, self._flat.total)
        return dx

    def _err_flag(self):
        new = ("err", (1,), torch.int32) not in self._scratch
        f = self._buf("err", (1,), torch.int32, zero=True)
        if new:
            ops.register_error_flag(f, IndexError, "ArcMarginProduct: label out of range [0, out_feature)")
        return f

    def check_labels(self):
        """Raise if any label seen since the last check was outside [0, out_feature) (reference: scatter_ raises); also reads
        every other registered device error flag (token / position ids of the text tower feeding this head)."""
        f = self._scratch.get(("err", (1,), torch.int32))
        if f is not None and int(f.item()) != 0:
            f.zero_()
            raise IndexError("ArcMarginProduct: label out of range [0, out_feature)")
        ops.check_device_flags()

    # ---- public API
    def forward(self, x, label):                                                     # arcface.py:45-63
        return _ArcLogitsFn.apply(x, self.weight, self, label)

    def forward_test(self, x):                                                       # arcface.py:65-67
        with torch.no_grad():
            cos, _ = self._cosines(x)
        return cos[:, :self.out_feature].clone()

    def forward_loss(self, x, label, want_argmax=True):
        """Fused margin + scale + mean cross-entropy.  Returns (loss scalar, argmax int64 [B])."""
        return _ArcLossFn.apply(x, self.weight, self, label)


class _ArcLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, mod, label):
        cos, st = mod._cosines(x)
        C = mod.out_feature
        label = label.contiguous()
        logits = cos.clone()
        ops.lib.arcface_margin(logits.data_ptr(), st["ldc"], label.data_ptr(), st["B"], C, mod.s, mod.m,
                               int(mod.easy_margin), mod._err_flag().data_ptr(), ops._stream())
        mod.check_labels()   # API path keeps the reference's eager error (costs one host sync)
        ctx.mod, ctx.st, ctx.label = mod, st, label
        ctx.cos = cos.clone()
        return logits[:, :C]

    @staticmethod
    def backward(ctx, dlogits):
        mod, st = ctx.mod, ctx.st
        mod._check_gen(st)
        C = mod.out_feature
        dlogits = dlogits.contiguous().float()
        dcos = mod._buf("dcos", (st["B"], st["ldc"]), torch.bfloat16)
        ops.lib.arcface_dlogits_to_dcos(dlogits.data_ptr(), dlogits.stride(0), ctx.cos.data_ptr(), st["ldc"],
                                        ctx.label.data_ptr(), dcos.data_ptr(), st["B"], C, mod.s, mod.m,
                                        int(mod.easy_margin), ops._stream())
        dx = mod._backward_from_dcos(st, dcos, ctx.cos)
        return dx, None, None, None


class _ArcLossFn(torch.autograd.Function):
    """loss = CrossEntropyLoss(ArcMarginProduct(x, label), label)  (arcface.py:45-63 + multimodal_classifier_train.py:188) and the argmax
    of the margin logits (:191), without a [B, C] tensor other than the cosines: written once by the cosine product (whose epilogue
    also leaves the softmax statistics), read once by the backward."""

    @staticmethod
    def forward(ctx, x, weight, mod, label):
        cos, st = mod._cosines(x, product=False)
        B, C, D, ldc = st["B"], mod.out_feature, mod.in_feature, st["ldc"]
        label = label.contiguous()
        dev = x.device
        loss_b = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        argmax = torch.empty(B, dtype=torch.int64, device=dev)
        rowst = mod._buf("rowst", (B, 4), torch.float32)
        part = mod._buf("part", (B * ((ldc + 63) // 64) * 4,), torch.float32)
        ops.lib.arcface_fwd_fused(st["xh"].data_ptr(), st["whp"].data_ptr(), cos.data_ptr(), ldc, label.data_ptr(), part.data_ptr(),
                                  part.numel(), rowst.data_ptr(), loss_b.data_ptr(), argmax.data_ptr(), loss.data_ptr(), B, C, D,
                                  mod.s, mod.m, int(mod.easy_margin), mod._err_flag().data_ptr(), ops._stream())
        ctx.mod, ctx.st, ctx.cos, ctx.label, ctx.rowst = mod, st, cos, label, rowst       # module scratch, intact until the next forward
        ctx.mark_non_differentiable(argmax)
        return loss, argmax

    @staticmethod
    def backward(ctx, dloss, _dargmax):
        mod, st = ctx.mod, ctx.st
        mod._check_gen(st)
        B, C, ldc = st["B"], mod.out_feature, st["ldc"]
        dloss = dloss.detach().reshape(1).float().contiguous()           # the upstream gradient stays on the device: the kernel reads it
        dcos = mod._buf("dcos", (B, ldc), torch.bfloat16)
        if ldc % 256 == 0 and _FUSED_DW:
            rowvec = mod._buf("rowvec", (2, C), torch.float32)
            ops.lib.arcface_dcos_rowfix(ctx.cos.data_ptr(), ldc, ctx.label.data_ptr(), ctx.rowst.data_ptr(), dloss.data_ptr(), 1.0 / B,
                                        st["inv_w"].data_ptr(), dcos.data_ptr(), rowvec.data_ptr(), B, C, mod.s, mod.m,
                                        int(mod.easy_margin), ops._stream())
        else:       # small heads (class dimension padded to 8, not 256): dcos from the row statistics, row vectors by their own pass
            rowvec = None
            lse = ctx.rowst[:, 0].contiguous()
            ops.lib.arcface_dcos_from_lse(ctx.cos.data_ptr(), ldc, ctx.label.data_ptr(), lse.data_ptr(),
                                          (dloss / B).expand(B).contiguous().data_ptr(), dcos.data_ptr(), B, C, 0, mod.s, mod.m,
                                          int(mod.easy_margin), ops._stream())
        dx = mod._backward_from_dcos(st, dcos, ctx.cos, rowvec=rowvec)
        return dx, None, None, None


class _GlueFn(torch.autograd.Function):
    """final = cat(normalize(img), normalize(txt), dim=1)   (multimodal_classifier.py:54-56)"""

    @staticmethod
    def forward(ctx, img, txt):
        img = img.contiguous().float()
        txt = txt.contiguous().float()
        B, Di = img.shape
        Dt = txt.shape[1]
        if Di % 4 or Dt % 4:
            raise ValueError("two-tower glue: embedding widths must be multiples of 4")
        final = torch.empty((B, Di + Dt), dtype=torch.float32, device=img.device)
        inv_i = torch.empty(B, dtype=torch.float32, device=img.device)
        inv_t = torch.empty(B, dtype=torch.float32, device=img.device)
        ops.l2norm_fwd(img, final, None, 0, inv_i)
        ops.l2norm_fwd(txt, final, None, Di, inv_t)
        ctx.save_for_backward(img, txt, inv_i, inv_t)
        return final

    @staticmethod
    def backward(ctx, dfinal):
        img, txt, inv_i, inv_t = ctx.saved_tensors
        dfinal = dfinal.contiguous().float()
        dimg = torch.empty_like(img)
        dtxt = torch.empty_like(txt)
        ops.l2norm_bwd(img, inv_i, dfinal, 0, dimg)
        ops.l2norm_bwd(txt, inv_t, dfinal, img.shape[1], dtxt)
        return dimg, dtxt


def glue_concat(img_emb, txt_emb):
    if not (img_emb.is_cuda and txt_emb.is_cuda):
        raise MmsimError("glue_concat: embeddings must be on the GPU; the HIP path has no CPU fallback")
    return _GlueFn.apply(img_emb, txt_emb)
