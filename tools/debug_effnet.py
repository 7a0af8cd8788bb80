"""Debug helper (GPU box): per-block comparison of the HIP image tower with the CPU oracle."""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import effnet_ref
from multimodalsimilar_amd.effnet import EfficientNet
warnings.simplefilter("ignore")
name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b0"
torch.manual_seed(0)
m = EfficientNet(name, seed=0)
sd = {"backbone." + k: v.detach().clone() for k, v in m.state_dict().items()}
m.to("cuda").train()
B, R = 8, 64
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 3, R, R, generator=g)
taps = {}
f = effnet_ref.backbone_forward(sd, name, x, training=True, taps=taps)
with torch.no_grad():
    st = m._run_forward(x.cuda())
def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
def to_nhwc(t):
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
print("stem", rel(st.x0, to_nhwc(taps["stem"])))
outs = [bs.x_in for bs in st.blocks[1:]] + [st.x_last]
for b, o in zip(m.arch.blocks, outs):
    print(b.name, b.type, b.cin, b.mid, b.cout, "k", b.k, "s", b.stride, "skip", b.skip, "relerr", round(rel(o, to_nhwc(taps[b.name])), 4))
print("pooled", rel(st.pooled, f.mean((2, 3))))

# ---- L2-relative errors incl. gradients through CvClassifier (use_fc=False)
from cv_classifier import CvClassifier
from oracle import arcface_ref
def l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()
for Bx, Rx in ((8, 64), (16, 96)):
    torch.manual_seed(0)
    model = CvClassifier(name, 64, 50, pretrained=False, use_fc=False)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to("cuda").train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(Bx, 3, Rx, Rx, generator=g)
    y = torch.randint(0, 50, (Bx,), generator=g)
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    emb_ref = effnet_ref.cv_predict_emb(sdr, name, x, use_fc=False, training=True)
    loss_ref = arcface_ref.ce_loss(arcface_ref.arcface_forward(emb_ref, sdr["classifier.weight"], y, 64.0, 0.2), y)
    loss_ref.backward()
    loss, _ = model.forward_loss(x.cuda(), y.cuda())
    loss.backward()
    emb = model.predict_emb(x.cuda())
    named = dict(model.named_parameters())
    gmax = max(v.grad.norm().item() for v in sdr.values() if torch.is_tensor(v) and v.grad is not None)
    errs = sorted([(l2(named[k].grad, sdr[k].grad), k, sdr[k].grad.norm().item()) for k in named
                   if sdr[k].grad is not None and sdr[k].grad.norm().item() > 1e-6 * gmax])
    n = len(errs)
    print(f"B={Bx} R={Rx}: emb L2 {l2(emb, emb_ref):.4f} max {rel(emb, emb_ref):.4f} loss {loss.item():.4f}/{loss_ref.item():.4f} "
          f"grad L2 median {errs[n//2][0]:.4f} p90 {errs[int(n*0.9)][0]:.4f} max {errs[-1][0]:.4f} ({errs[-1][1]}, norm {errs[-1][2]:.2e})")
    print("   worst5:", [(round(e, 3), k, f"{nn:.1e}") for e, k, nn in errs[-5:]])
