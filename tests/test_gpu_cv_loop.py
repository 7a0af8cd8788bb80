"""The image-only training loop on the MI355X (cv_classifier_train_daodian.py:108-142, 264-267, 292): fused Adam against
torch.optim.Adam, the per-epoch margin annealing changing the logits through a kernel ARGUMENT (no rebuild), loss going down."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_fused_adam_equals_torch_adam():
    from multimodalsimilar_amd.optim import FusedAdam
    from multimodalsimilar_amd.head import ArcMarginProduct
    torch.manual_seed(0)
    head = ArcMarginProduct(64, 40).to(DEV)
    ref_p = torch.nn.Parameter(head.weight.detach().clone())
    opt, ref = FusedAdam(head, lr=1e-3), torch.optim.Adam([ref_p], lr=1e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=3, eta_min=1e-6)
    rsched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(ref, T_0=3, eta_min=1e-6)
    g = torch.Generator(device=DEV).manual_seed(1)
    for step in range(7):
        grad = torch.randn(40, 64, device=DEV, generator=g) * (10.0 ** (step % 3 - 2))
        head._bind_grads()
        head.weight.grad.copy_(grad)
        ref_p.grad = grad.clone()
        opt.step(); ref.step()
        sched.step(); rsched.step()
        assert torch.allclose(head.weight.detach(), ref_p.detach(), rtol=2e-6, atol=2e-7), step


def test_cv_loop_trains_and_anneals_the_margin_without_rebuilding():
    import warnings
    from oracle import arcface_ref
    from multimodalsimilar_amd import train as T
    from multimodalsimilar_amd import build as B_
    warnings.simplefilter("ignore")
    cfg = dict(kind="cv", image="efficientnet_b0", res=64, batch=16, classes=50, fc_dim=64, use_fc=True)
    model = T.build_model(cfg, DEV, seed=0)
    model.dropout.p = 0.0
    loop = T.CvTrainLoop(model)
    stamp = open(B_.STAMP).read()
    batch = T.synthetic_batch(cfg, DEV, seed=2)
    losses, margins = [], []
    for epoch in range(3):
        for it in range(3):
            loss, pred = loop.step(batch)
            losses.append(float(loss.item()))
        margins.append(model.classifier.m)
        # the literal API path sees the CURRENT margin: logits == oracle at that margin on the same embedding
        model.eval()
        with torch.no_grad():
            emb = model.predict_emb(batch["img_tensor"])
        lg = model.classifier(emb, batch["labels"])
        ref = arcface_ref.arcface_forward(emb.cpu(), model.classifier.weight.detach().cpu(), batch["labels"].cpu(), 64.0, model.classifier.m)
        assert (lg.cpu() - ref).abs().max() < 0.4, epoch
        vloss, vpred = loop.evaluate(batch)
        assert torch.isfinite(vloss)
        loop.end_epoch()
    assert [round(m, 2) for m in margins] == [0.2, 0.24, 0.28]            # CvClassifier default m = 0.2, +0.04 per epoch
    assert all(torch.isfinite(torch.tensor(losses))) and min(losses[3:]) < losses[0]
    assert open(B_.STAMP).read() == stamp                                  # same binary throughout: the margin is an argument
    assert abs(loop.optimizer.param_groups[0]["lr"] - (1e-6 + (1e-3 - 1e-6) * (1 + __import__("math").cos(__import__("math").pi * 3 / 7)) / 2)) < 1e-12
