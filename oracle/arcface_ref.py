"""Oracle: ArcFace additive-angular-margin head, fp32 CPU.  Test infrastructure only.

Follows /root/reference/arcface.py:
  constants   arcface.py:27-33  (cos_m, sin_m, th = cos(pi-m), mm = sin(pi-m)*m)
  forward     arcface.py:45-63
  forward_test arcface.py:65-67
  update_m    arcface.py:35-42
Loss: nn.CrossEntropyLoss() mean over batch (multimodal_classifier_train.py:167,188).
The analytic backward restates SURVEY.md Appendix A and is cross-checked against autograd
in tests/test_oracle_arcface.py.
"""
import math
import torch
import torch.nn.functional as F


def margin_constants(m):
    return dict(cos_m=math.cos(m), sin_m=math.sin(m), th=math.cos(math.pi - m),
                mm=math.sin(math.pi - m) * m)


def update_m(m, delta):
    """arcface.py:35-42 -- returns the new margin (unchanged if out of [1e-6, 1.0])."""
    u = m + delta
    return u if (u >= 1e-6 and u <= 1.0) else m


def arcface_forward(x, weight, label, s=64.0, m=0.40, easy_margin=False):
    """arcface.py:45-63.  x [B,D] fp32, weight [C,D] fp32, label [B] int64 -> logits [B,C]."""
    k = margin_constants(m)
    cosine = F.linear(F.normalize(x), F.normalize(weight))                     # :47
    sine = torch.sqrt(1.0 - torch.pow(cosine, 2))                              # :49
    phi = cosine * k["cos_m"] - sine * k["sin_m"]                              # :50
    if easy_margin:
        phi = torch.where(cosine > 0, phi, cosine)                             # :53
    else:
        phi = torch.where((cosine - k["th"]) > 0, phi, cosine - k["mm"])       # :55
    one_hot = torch.zeros_like(cosine)
    one_hot.scatter_(1, label.view(-1, 1), 1)                                  # :58-59
    out = (one_hot * phi) + ((1.0 - one_hot) * cosine)                         # :60
    return out * s                                                             # :61


def arcface_forward_test(x, weight):
    """arcface.py:65-67 -- cosine logits, no margin, no scale."""
    return F.linear(F.normalize(x), F.normalize(weight))


def ce_loss(logits, label):
    """multimodal_classifier_train.py:167,188 -- mean cross entropy."""
    return F.cross_entropy(logits, label)


def arcface_ce_analytic(x, weight, label, s=64.0, m=0.40, easy_margin=False):
    """Closed-form loss, dX, dW without autograd.

    Returns (loss, logits, dx, dw, argmax)."""
    k = margin_constants(m)
    B = x.shape[0]
    xn = x.norm(dim=1, keepdim=True).clamp_min(1e-12)
    wn = weight.norm(dim=1, keepdim=True).clamp_min(1e-12)
    xh, wh = x / xn, weight / wn
    cos = xh @ wh.t()
    idx = torch.arange(B)
    ct = cos[idx, label]
    sine = torch.sqrt(1.0 - ct * ct)
    phi = ct * k["cos_m"] - sine * k["sin_m"]
    if easy_margin:
        take = ct > 0
        alt = ct
    else:
        take = (ct - k["th"]) > 0
        alt = ct - k["mm"]
    zt = torch.where(take, phi, alt)
    z = cos.clone()
    z[idx, label] = zt
    z = z * s
    lse = torch.logsumexp(z, dim=1)
    loss = (lse - z[idx, label]).mean()
    g = torch.softmax(z, dim=1)
    g[idx, label] -= 1.0
    g = g / B
    dcos = g * s
    slope = torch.where(take, k["cos_m"] + k["sin_m"] * ct / sine, torch.ones_like(ct))
    dcos[idx, label] = g[idx, label] * s * slope
    dxh = dcos @ wh
    dwh = dcos.t() @ xh
    dx = (dxh - xh * (xh * dxh).sum(1, keepdim=True)) / xn
    dw = (dwh - wh * (wh * dwh).sum(1, keepdim=True)) / wn
    return loss, z, dx, dw, z.argmax(dim=1)


def glue_concat(img_emb, txt_emb):
    """multimodal_classifier.py:54-56 -- L2-normalise each tower embedding, concatenate."""
    return torch.cat((F.normalize(img_emb, p=2, dim=1), F.normalize(txt_emb, p=2, dim=1)), 1)
