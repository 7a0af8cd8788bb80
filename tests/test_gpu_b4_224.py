"""EfficientNet-B4 at its PRODUCTION geometry (224 x 224: the 112^2 / 56^2 / 28^2 stages where the streaming 1x1-conv kernels and the
large depthwise tile shapes are selected) against the CPU oracle -- VERDICT r3 item 3.  cv_classifier.py:47-55 over timm
`efficientnet_b4`; oracle/effnet_ref.py restates it (parity unpinned w.r.t. timm).

Rounds 1-3 compared the whole tower with the oracle at 64^2 / 96^2 only; at 224^2 there were finite / non-zero property checks.  Here:
  (a) conditioned weights (what training leaves: tests/test_gpu_eval_parity.py), B = 8, eval AND train mode, embedding and loss at
      north_star's 1e-2, with the kernels that ran COUNTED: the streaming pw_* kernels and the tiled depthwise kernels must be the ones
      the 112^2 / 56^2 stages took (the `*_eligible` predicates the host code itself consults, and call counters on the C ABI);
  (b) RANDOM-INIT weights in train mode -- the weights bench.py runs -- at 224^2 with B = 32 (1 568 samples per channel at the 7^2
      stage instead of 64 at 64^2): the embedding error measured at the shape that matters, asserted at 1e-2 if the hardware meets it
      (the bound used at 64^2 was the emulation-relative 2 d0 + 1 %).
"""
import time
import warnings

import pytest
import torch

from parity_log import check, record
from test_gpu_eval_parity import conditioned_cv, l2err, NORTH_STAR

pytestmark = pytest.mark.gpu
DEV = "cuda"
NAME = "efficientnet_b4"
# entry points whose use at 224^2 is asserted (forward, then backward)
STREAMING = ("pw_expand_fwd", "pw_project_fwd_xf", "pw_project_fwd", "dwtile_fwd", "dw5m_fwd", "pw_project_bwd_xf", "pw_expand_bwd", "dwtile_bwd",
             "dw5m_bwd", "dwtile_bwd_s2")


class _Counter:
    """Counts calls of chosen C-ABI entry points (the binding caches one callable per entry point on the lib object)."""

    def __init__(self, names):
        from multimodalsimilar_amd._lib import lib
        self.lib, self.names, self.n, self.orig = lib, names, {k: 0 for k in names}, {}

    def __enter__(self):
        for k in self.names:
            fn = getattr(self.lib, k)
            self.orig[k] = fn

            def wrapped(*a, _k=k, _fn=fn):
                self.n[_k] += 1
                return _fn(*a)
            object.__setattr__(self.lib, k, wrapped)
        return self

    def __exit__(self, *exc):
        for k, fn in self.orig.items():
            object.__setattr__(self.lib, k, fn)


def _stage_shapes_take_the_streaming_kernels(B):
    """The predicates effnet.py consults, evaluated on B4's 224^2 stage shapes."""
    from multimodalsimilar_amd._lib import lib
    P112, P56, P28 = B * 112 * 112, B * 56 * 56, B * 28 * 28
    assert lib.pw_project_fwd_eligible(P112, 112 * 112, 48, 24) and lib.pw_project_bwd_eligible(P112, 112 * 112, 48, 24)      # stage 0 (ds)
    assert lib.pw_expand_fwd_eligible(P56, 192, 32) and lib.pw_expand_bwd_eligible(P56, 192, 32)                               # stage 1 expansion
    assert lib.pw_project_fwd_eligible(P56, 56 * 56, 192, 32) and lib.pw_project_bwd_eligible(P56, 56 * 56, 192, 32)          # stage 1 projection
    assert lib.pw_expand_bwd_eligible(P112, 144, 24)                                                                           # stage 1 block 0 (112^2 in)
    assert lib.pw_expand_bwd_eligible(P28, 336, 56)                                                                            # stage 2 (28^2, 8-wave form)
    assert lib.pw_project_fwd_eligible(P28, 28 * 28, 192, 56)                                                                  # first 28^2 block


def test_b4_at_224_conditioned_eval_and_train_match_the_oracle():
    from oracle import effnet_ref, arcface_ref
    from multimodalsimilar_amd import ops
    B = 8
    _stage_shapes_take_the_streaming_kernels(B)
    t0 = time.time()
    model, sd, x = conditioned_cv(NAME, False, seed=11, res=224, batch=B)
    g = torch.Generator().manual_seed(12)
    xe = torch.randn(B, 3, 224, 224, generator=g)
    y = torch.randint(0, 50, (B,), generator=g)
    # ---- eval mode (running statistics), fresh data
    ref = effnet_ref.cv_predict_emb(sd, NAME, xe, use_fc=False, training=False)
    model.to(DEV).eval()
    with _Counter(STREAMING) as cnt, torch.no_grad():
        emb = model.predict_emb(xe.to(DEV))
    e_eval = l2err(emb, ref)
    # 32 depthwise layers: the tiled VALU kernels (3 x 3 and the stride-2 blocks) + the matrix-core kernels (the 16 5 x 5 stride-1 blocks)
    assert cnt.n["pw_expand_fwd"] >= 3 and cnt.n["pw_project_fwd_xf"] + cnt.n["pw_project_fwd"] >= 6, cnt.n
    assert cnt.n["dwtile_fwd"] + cnt.n["dw5m_fwd"] >= 28 and cnt.n["dw5m_fwd"] == 16, cnt.n
    # ---- train mode (batch statistics), forward + backward
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    emb_ref = effnet_ref.cv_predict_emb(sdr, NAME, x, use_fc=False, training=True)
    loss_ref = arcface_ref.ce_loss(arcface_ref.arcface_forward(emb_ref, sdr["classifier.weight"], y, 64.0, 0.2), y)
    loss_ref.backward()
    grads = {k: v.grad for k, v in sdr.items() if torch.is_tensor(v) and v.grad is not None}
    t_oracle = time.time() - t0
    ops.set_deterministic(True)
    try:
        model.train()
        with _Counter(STREAMING) as cnt:
            emb_t = model.predict_emb(x.to(DEV))
            loss, _ = model.forward_loss(x.to(DEV), y.to(DEV))
            loss.backward()
            torch.cuda.synchronize()
    finally:
        ops.set_deterministic(False)
    # the 112^2 / 56^2 stages ran the streaming kernels and the tiled depthwise kernels, forward and backward (two forwards here)
    assert cnt.n["pw_expand_fwd"] >= 6 and cnt.n["pw_project_fwd_xf"] + cnt.n["pw_project_fwd"] >= 12, cnt.n
    assert cnt.n["pw_project_bwd_xf"] >= 5 and cnt.n["pw_expand_bwd"] >= 8, cnt.n
    assert cnt.n["dwtile_bwd"] + cnt.n["dw5m_bwd"] >= 28 and cnt.n["dwtile_fwd"] + cnt.n["dw5m_fwd"] >= 56 and cnt.n["dw5m_fwd"] == 32, cnt.n
    assert cnt.n["dwtile_bwd_s2"] == 4, cnt.n          # the first block of the four stride-2 stages: the fused kernel, not the round-1 trio
    assert cnt.n["dw5m_bwd"] == 9, cnt.n               # the 28^2 and 14^2 5 x 5 blocks (3 + 6); the 7^2 ones stay on the VALU kernel (measured faster there)
    named = dict(model.named_parameters())
    gmax = max(v.norm().item() for v in grads.values())
    keys = [k for k in named if k in grads and named[k].grad is not None and grads[k].norm().item() > 1e-4 * gmax]
    ge = sorted(l2err(named[k].grad, grads[k]) for k in keys)
    e_train = l2err(emb_t, emb_ref.detach())
    le = abs(loss.item() - loss_ref.item()) / loss_ref.item()
    print(f"\n[B4 @224^2, B={B}, conditioned] eval emb L2 err {e_eval:.4f}; train emb {e_train:.4f}, loss {loss.item():.4f} vs {loss_ref.item():.4f} "
          f"({le:.2e}); grad L2 median {ge[len(ge) // 2]:.3f} / p90 {ge[int(0.9 * len(ge))]:.3f} over {len(keys)} tensors; oracle {t_oracle:.0f} s; "
          f"kernel calls {cnt.n}")
    tag = "b4_at_224[conditioned]"
    check(tag, "eval-mode embedding relative L2", e_eval, NORTH_STAR)
    check(tag, "train-mode embedding relative L2", e_train, NORTH_STAR)
    check(tag, "loss relative error", le, NORTH_STAR)
    check(tag, "median parameter-gradient relative L2", ge[len(ge) // 2], 3e-2)
    check(tag, "p90 parameter-gradient relative L2", ge[int(0.9 * len(ge))], 4e-2)


def test_b4_at_224_random_init_train_mode():
    """The weights bench.py runs (random init, batch-statistic BatchNorm) at the production geometry, forward and backward.  At 64^2 /
    B = 16 this case measured 2.29 % on the embedding and 5.8 % on the median parameter gradient (tests/test_gpu_image_tower.py: 2x2 ...
    4x4 maps give 64-256 samples per channel and ~100 train-mode BatchNorms amplify every rounding); here every late-stage BatchNorm
    sees 32 x 49 = 1 568 samples."""
    from oracle import effnet_ref, arcface_ref
    from cv_classifier import CvClassifier
    from multimodalsimilar_amd import ops
    warnings.simplefilter("ignore")
    B = 32
    torch.manual_seed(0)
    model = CvClassifier(NAME, 64, 50, pretrained=False, use_fc=False)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, 224, 224, generator=g)
    y = torch.randint(0, 50, (B,), generator=g)
    t0 = time.time()
    with torch.no_grad():
        emu = effnet_ref.cv_predict_emb(sd, NAME, x, use_fc=False, training=True, emulate="fp16")
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    ref = effnet_ref.cv_predict_emb(sdr, NAME, x, use_fc=False, training=True)
    loss_ref = arcface_ref.ce_loss(arcface_ref.arcface_forward(ref, sdr["classifier.weight"], y, 64.0, 0.2), y)
    loss_ref.backward()
    grads = {k: v.grad for k, v in sdr.items() if torch.is_tensor(v) and v.grad is not None}
    ref = ref.detach()
    t_oracle = time.time() - t0
    ops.set_deterministic(True)
    try:
        model.to(DEV).train()
        emb = model.predict_emb(x.to(DEV))
        loss, _ = model.forward_loss(x.to(DEV), y.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ops.set_deterministic(False)
    named = dict(model.named_parameters())
    gmax = max(v.norm().item() for v in grads.values())
    keys = [k for k in named if k in grads and named[k].grad is not None and grads[k].norm().item() > 1e-5 * gmax]
    ge = sorted(l2err(named[k].grad, grads[k]) for k in keys)
    e, d0 = l2err(emb, ref), l2err(emu, ref)
    le = abs(loss.item() - loss_ref.item()) / loss_ref.item()
    print(f"\n[B4 @224^2, B={B}, random init, train mode] emb L2 err {e:.4f} (oracle under fp16-storage emulation d0 {d0:.4f}); loss {loss.item():.4f} vs "
          f"{loss_ref.item():.4f} ({le:.2e}); grad L2 median {ge[len(ge) // 2]:.3f} / p90 {ge[int(0.9 * len(ge))]:.3f} over {len(keys)} tensors; oracle {t_oracle:.0f} s")
    tag = "b4_at_224[random-init train mode]"
    # MEASURED (r04, MI355X): embedding 1.08 % against the fp32 oracle -- the oracle ITSELF moves by d0 = 1.10 % when nothing but its stored
    # tensors are rounded to fp16, so the HIP path sits on the storage floor, and that floor is above north_star's 1e-2 for RANDOM-INIT
    # weights in train mode even at the production geometry (64^2 / B = 16 measured 2.29 %): ~100 batch-statistic BatchNorms over
    # untrained weights amplify each rounding.  The loss (what training sees) is 1.8e-4 off.  Asserted: a plain absolute bound with a
    # 15 % margin over the measurement, NOT 1e-2 and not an emulation-relative bound; DESIGN.md section 5 states the number.  The 1e-2
    # assertions on conditioned (trained-looking) weights at this geometry are in the test above: 0.26-0.27 %.
    record(tag, "oracle under fp16-storage emulation: embedding relative L2 (d0)", d0, 1.25e-2)
    check(tag, "embedding relative L2 (random init: above north_star's 1e-2, see comment)", e, 1.25e-2)
    check(tag, "loss relative error", le, NORTH_STAR)
    check(tag, "median parameter-gradient relative L2", ge[len(ge) // 2], 5e-2)
