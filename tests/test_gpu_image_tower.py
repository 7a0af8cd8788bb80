"""Image-tower parity on the MI355X: conv kernels against fp32 torch references of the same op, and the whole
EfficientNet / CvClassifier forward + backward against the CPU oracle (oracle/effnet_ref.py; PARITY UNPINNED
w.r.t. timm, see its header).  Element types (include/mmsim_hip.h, EfficientNet section): FORWARD tensors fp16 (`nhwc`, `.half()`),
GRADIENT tensors bf16 (`nhwc_g`, `.bfloat16()`): fp16-class tolerances on forward outputs (2e-3 of the output scale), 1e-2-class on
bf16 gradients, looser on deep-chain gradients."""
import pytest
import torch
import torch.nn.functional as F

from parity_log import check, record

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def rnd(*s, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(DEV)


_SCR = {}


def _lib():
    from multimodalsimilar_amd import ops
    from multimodalsimilar_amd._lib import lib
    if "t" not in _SCR:
        _SCR["t"] = torch.empty(8 << 20, device=DEV)
    return lib, ops._stream()


def scr():
    return _SCR["t"].data_ptr(), _SCR["t"].numel()


def nhwc(x):      # [B,C,H,W] fp32 -> [B*H*W, C] fp16: a FORWARD tensor of the image tower
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().half()


def nhwc_g(x):    # the same as bf16: a GRADIENT tensor
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().bfloat16()


FWD_TOL = 2e-3    # fp16 outputs: half an ulp is 2.4e-4 of the value; max-norm relative bound with margin for the accumulation order


def nchw(x, B, H, W):
    return x.float().view(B, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("K,S,H,C", [(3, 1, 12, 48), (5, 1, 10, 40), (3, 2, 16, 24), (5, 2, 14, 2688), (3, 1, 7, 272)])
def test_depthwise_conv_fwd_bwd(K, S, H, C):
    lib, s = _lib()
    B, W = 3, H + 2
    x = rnd(B, C, H, W, seed=1)
    w = rnd(C, 1, K, K, seed=2, scale=0.3)
    a = nhwc(x)
    xr = nchw(a, B, H, W).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=S, padding=K // 2, groups=C)
    Ho, Wo = ref.shape[2], ref.shape[3]
    wT = torch.empty(K * K, C, device=DEV)
    lib.dw_weight_to_tap_major(w.data_ptr(), wT.data_ptr(), C, K, s)
    z = torch.empty(B * Ho * Wo, C, dtype=torch.float16, device=DEV)
    sums = torch.zeros(2 * C, device=DEV)
    lib.dwconv_fwd(a.data_ptr(), wT.data_ptr(), z.data_ptr(), sums.data_ptr(), B, H, W, C, K, S, *scr(), s)
    assert relerr(nchw(z, B, Ho, Wo), ref) < FWD_TOL
    zf = z.float()
    assert relerr(sums[:C], zf.sum(0)) < 1e-3 and relerr(sums[C:], (zf * zf).sum(0)) < 1e-3
    dz = rnd(B, C, Ho, Wo, seed=3)
    dzb = nhwc_g(dz)
    ref.backward(nchw(dzb, B, Ho, Wo))
    # plain transposed conv (+ residual)
    res = rnd(B * H * W, C, seed=4).bfloat16()
    dx = torch.empty(B * H * W, C, dtype=torch.bfloat16, device=DEV)
    lib.dwconv_bwd_data(dzb.data_ptr(), wT.data_ptr(), None, None, None, None, None, res.data_ptr(), dx.data_ptr(), None,
                        B, H, W, C, K, S, *scr(), s)
    assert relerr(nchw(dx, B, H, W), xr.grad + nchw(res, B, H, W)) < 1e-2
    # fused with the producer's BN + SiLU backward
    z1 = rnd(B * H * W, C, seed=5).half()
    mean, rstd = rnd(C, seed=6, scale=0.1), 1 + 0.1 * rnd(C, seed=7).abs()
    scale, shift = 1 + 0.1 * rnd(C, seed=8), 0.1 * rnd(C, seed=9)
    bsum = torch.zeros(2 * C, device=DEV)
    lib.dwconv_bwd_data(dzb.data_ptr(), wT.data_ptr(), z1.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(),
                        shift.data_ptr(), None, dx.data_ptr(), bsum.data_ptr(), B, H, W, C, K, S, *scr(), s)
    y = (z1.float() * scale + shift).requires_grad_(True)
    F.silu(y).backward(torch.ones_like(y))
    dpre = xr.grad.permute(0, 2, 3, 1).reshape(-1, C) * y.grad
    assert relerr(dx, dpre) < 1e-2
    dxf = dx.float()
    assert relerr(bsum[:C], dxf.sum(0)) < 2e-3
    assert relerr(bsum[C:], (dxf * (z1.float() - mean) * rstd).sum(0)) < 2e-3
    # weight gradient
    gT = torch.zeros(K * K, C, device=DEV)
    lib.dwconv_bwd_weight(dzb.data_ptr(), a.data_ptr(), gT.data_ptr(), B, H, W, C, K, S, *scr(), s)
    g = torch.ones(C, 1, K, K, device=DEV)
    lib.dw_grad_from_tap_major(gT.data_ptr(), g.data_ptr(), C, K, s)
    assert relerr(g - 1, wr.grad) < 1e-2
    # the same gradient with the operand re-formed from the pre-BatchNorm tensor (stride-2 blocks store no a1): a = silu(sc z + sh)
    a_from = (F.silu(z1.float() * scale + shift)).half()
    gT_a = torch.zeros(K * K, C, device=DEV)
    lib.dwconv_bwd_weight(dzb.data_ptr(), a_from.data_ptr(), gT_a.data_ptr(), B, H, W, C, K, S, *scr(), s)
    gT_x = torch.zeros(K * K, C, device=DEV)
    lib.dwconv_bwd_weight_xf(dzb.data_ptr(), z1.data_ptr(), scale.data_ptr(), shift.data_ptr(), gT_x.data_ptr(), B, H, W, C, K, S, *scr(), s)
    assert relerr(gT_x, gT_a) < 2e-3


@pytest.mark.parametrize("K,S,H,C,xf", [(3, 1, 12, 48, True), (5, 1, 10, 40, True), (3, 2, 16, 144, False), (5, 2, 14, 192, False),
                                          (3, 1, 7, 272, True), (5, 1, 14, 960, True), (3, 1, 30, 24, False), (5, 1, 28, 336, True),
                                          (3, 2, 18, 1632, True)])
def test_tiled_depthwise_forward(K, S, H, C, xf):
    """LDS-tiled depthwise forward (csrc/mbconv.hip) with the producer's BN + SiLU applied while the tile is staged, against
    fp32 torch: conv of the ROUNDED activation, zero padding after the activation, statistics of the rounded outputs."""
    lib, s = _lib()
    B, W = 3, H + 3
    z1 = nhwc(rnd(B, C, H, W, seed=1))
    scale, shift = 1 + 0.1 * rnd(C, seed=8), 0.1 * rnd(C, seed=9)
    w = rnd(C, 1, K, K, seed=2, scale=0.3)
    a = F.silu(z1.float() * scale + shift).half() if xf else z1
    ref = F.conv2d(nchw(a, B, H, W), w, None, stride=S, padding=K // 2, groups=C)
    Ho, Wo = ref.shape[2], ref.shape[3]
    wT = torch.empty(K * K, C, device=DEV)
    lib.dw_weight_to_tap_major(w.data_ptr(), wT.data_ptr(), C, K, s)
    z = torch.full((B * Ho * Wo, C), float("nan"), dtype=torch.float16, device=DEV)
    sums = torch.zeros(2 * C, device=DEV)
    lib.dwtile_fwd(z1.data_ptr(), scale.data_ptr() if xf else None, shift.data_ptr() if xf else None, wT.data_ptr(), z.data_ptr(),
                   sums.data_ptr(), B, H, W, C, K, S, *scr(), s)
    assert relerr(nchw(z, B, Ho, Wo), ref) < FWD_TOL
    zf = z.float()
    assert relerr(sums[:C], zf.sum(0)) < 1e-3 and relerr(sums[C:], (zf * zf).sum(0)) < 1e-3


@pytest.mark.parametrize("K,H,C,plain,resid", [(3, 12, 48, False, False), (5, 10, 40, False, False), (3, 7, 272, False, False),
                                               (5, 14, 960, False, False), (3, 30, 24, True, True), (3, 20, 48, True, False),
                                               (5, 28, 336, False, False), (5, 7, 1632, False, False), (3, 56, 192, False, False)])
def test_tiled_depthwise_backward_equals_the_unfused_kernels(K, H, C, plain, resid):
    """mmsim_dwtile_bwd (one kernel: depthwise-BN + SiLU + gate backward, data + weight gradients, expand-BN + SiLU backward
    on the way out) against the sequence it replaces -- bn_bwd, dwconv_bwd_weight, dwconv_bwd_data -- whose members are checked
    against fp32 torch autograd above."""
    lib, s = _lib()
    B, W = 3, H + 1
    P, HW = B * H * W, H * W
    z2 = rnd(P, C, seed=1, scale=1.5).half()
    dy = rnd(P, C, seed=2).bfloat16()
    z1 = rnd(P, C, seed=3, scale=1.5).half()
    w = rnd(C, 1, K, K, seed=4, scale=0.3)
    gate, dsq = torch.sigmoid(rnd(B, C, seed=5)), rnd(B, C, seed=6)
    mk = lambda sd: (rnd(C, seed=sd, scale=0.1), 1 + 0.1 * rnd(C, seed=sd + 1).abs(), 1 + 0.1 * rnd(C, seed=sd + 2), 0.1 * rnd(C, seed=sd + 3))
    mu2, rs2, sc2, sh2 = mk(10)
    mu1, rs1, sc1, sh1 = mk(20)
    res = rnd(P, C, seed=7).bfloat16() if resid else None
    wT = torch.empty(K * K, C, device=DEV)
    lib.dw_weight_to_tap_major(w.data_ptr(), wT.data_ptr(), C, K, s)
    # ---- the unfused sequence
    sums2 = torch.zeros(2 * C, device=DEV)
    dz2 = torch.empty(P, C, dtype=torch.bfloat16, device=DEV)
    dg_ref, db_ref = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    lib.bn_bwd(dy.data_ptr(), z2.data_ptr(), mu2.data_ptr(), rs2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), gate.data_ptr(),
               dsq.data_ptr(), HW, 1, sums2.data_ptr(), 0, dz2.data_ptr(), dg_ref.data_ptr(), db_ref.data_ptr(), P, C, *scr(), s)
    a1 = z1 if plain else F.silu(z1.float() * sc1 + sh1).half()
    gT_ref = torch.zeros(K * K, C, device=DEV)
    lib.dwconv_bwd_weight(dz2.data_ptr(), a1.data_ptr(), gT_ref.data_ptr(), B, H, W, C, K, 1, *scr(), s)
    out_ref = torch.empty(P, C, dtype=torch.bfloat16, device=DEV)
    sums1_ref = torch.zeros(2 * C, device=DEV)
    if plain:
        lib.dwconv_bwd_data(dz2.data_ptr(), wT.data_ptr(), None, None, None, None, None, res.data_ptr() if resid else None,
                            out_ref.data_ptr(), None, B, H, W, C, K, 1, *scr(), s)
    else:
        lib.dwconv_bwd_data(dz2.data_ptr(), wT.data_ptr(), z1.data_ptr(), mu1.data_ptr(), rs1.data_ptr(), sc1.data_ptr(), sh1.data_ptr(),
                            None, out_ref.data_ptr(), sums1_ref.data_ptr(), B, H, W, C, K, 1, *scr(), s)
    # ---- the fused kernel
    out = torch.full((P, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    sums1, gT = torch.zeros(2 * C, device=DEV), torch.zeros(K * K, C, device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    n = lambda t: None if t is None else t.data_ptr()
    lib.dwtile_bwd(dy.data_ptr(), z2.data_ptr(), sc2.data_ptr(), sh2.data_ptr(), mu2.data_ptr(), rs2.data_ptr(), sums2.data_ptr(),
                   gate.data_ptr(), dsq.data_ptr(), z1.data_ptr(), *(4 * [None] if plain else [sc1.data_ptr(), sh1.data_ptr(), mu1.data_ptr(), rs1.data_ptr()]),
                   n(res), wT.data_ptr(), out.data_ptr(), None if plain else sums1.data_ptr(), gT.data_ptr(), dg.data_ptr(), db.data_ptr(),
                   B, H, W, C, K, *scr(), s)
    assert relerr(out, out_ref) < 1e-2
    assert relerr(gT, gT_ref) < 5e-3
    assert relerr(dg, dg_ref) < 1e-5 and relerr(db, db_ref) < 1e-5
    if not plain:
        assert relerr(sums1, sums1_ref) < 5e-3


@pytest.mark.parametrize("P,mid,cin,skip", [(1280, 192, 32, True), (6400, 144, 24, False), (832, 336, 56, True), (64, 48, 8, False),
                                             (2048, 192, 32, False)])
def test_fused_expand_backward_equals_bn_backward_plus_two_products(P, mid, cin, skip):
    """mmsim_pw_expand_bwd (BatchNorm backward on the way into LDS, dx and dW1 from the same staged strip) against the sequence
    it replaces: mmsim_bn_bwd (sums given) + the dgrad and wgrad products."""
    from multimodalsimilar_amd import ops
    lib, s = _lib()
    assert lib.pw_expand_bwd_eligible(P, mid, cin) == 1
    dpre = rnd(P, mid, seed=1).bfloat16()
    z1 = rnd(P, mid, seed=2, scale=1.5).half()
    x = rnd(P, cin, seed=3).half()
    res = rnd(P, cin, seed=4).bfloat16() if skip else None
    w1 = rnd(mid, cin, seed=5, scale=0.2).bfloat16()
    mean, rstd, scale = rnd(mid, seed=6, scale=0.1), 1 + 0.1 * rnd(mid, seed=7).abs(), 1 + 0.1 * rnd(mid, seed=8)
    shift = 0.1 * rnd(mid, seed=9)
    # the BatchNorm-backward sums of dpre (what mmsim_dwtile_bwd / mmsim_dwconv_bwd_data leave)
    df, zh = dpre.float(), (z1.float() - mean) * rstd
    sums = torch.cat([df.sum(0), (df * zh).sum(0)]).contiguous()
    # ---- unfused
    dz1 = torch.empty(P, mid, dtype=torch.bfloat16, device=DEV)
    dg_ref, db_ref = torch.zeros(mid, device=DEV), torch.zeros(mid, device=DEV)
    lib.bn_bwd(dpre.data_ptr(), z1.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, None, 1, 0,
               sums.data_ptr(), 1, dz1.data_ptr(), dg_ref.data_ptr(), db_ref.data_ptr(), P, mid, *scr(), s)
    gw_ref = torch.full((mid, cin), 0.5, device=DEV)
    ops.gemm(dz1, x, gw_ref, trans_a=True, b_kmajor=False, accumulate=True)
    dx_ref = torch.empty(P, cin, dtype=torch.bfloat16, device=DEV)
    ops.gemm(dz1, w1, dx_ref, b_kmajor=False, epilogue=ops.EPI_ADD if skip else ops.EPI_NONE, aux_in=res)
    # ---- fused
    gw = torch.full((mid, cin), 0.5, device=DEV)
    dx = torch.full((P, cin), float("nan"), dtype=torch.bfloat16, device=DEV)
    dg, db = torch.zeros(mid, device=DEV), torch.zeros(mid, device=DEV)
    lib.pw_expand_bwd(dpre.data_ptr(), z1.data_ptr(), x.data_ptr(), None if res is None else res.data_ptr(), w1.data_ptr(),
                      scale.data_ptr(), mean.data_ptr(), rstd.data_ptr(), sums.data_ptr(), dx.data_ptr(), gw.data_ptr(), dg.data_ptr(),
                      db.data_ptr(), P, mid, cin, *scr(), s)
    assert relerr(dx, dx_ref) < 1e-2
    assert relerr(gw - 0.5, gw_ref - 0.5) < 5e-3
    assert relerr(dg, dg_ref) < 1e-5 and relerr(db, db_ref) < 1e-5
    # and against fp32 autograd of the same formulae (both paths round dz1 to bf16; the weight gradient also rounds x to bf16)
    dz = scale * (df - sums[:mid] / P - zh * sums[mid:] / P)
    assert relerr(dx, dz @ w1.float() + (res.float() if skip else 0)) < 1.5e-2
    assert relerr(gw - 0.5, dz.t() @ x.float()) < 1e-2


@pytest.mark.parametrize("C,P", [(24, 1000), (336, 777), (2688, 98)])
def test_batchnorm_stats_apply_backward(C, P):
    lib, s = _lib()
    HW = 7 if P % 7 == 0 else 1
    B = P // HW
    z = rnd(P, C, seed=1, scale=2.0).half()
    gamma, beta = 1 + 0.2 * rnd(C, seed=2), 0.2 * rnd(C, seed=3)
    sums = torch.zeros(2 * C, device=DEV)
    lib.bn_stats(z.data_ptr(), sums.data_ptr(), P, C, *scr(), s)
    mean, rstd, scale, shift = (torch.empty(C, device=DEV) for _ in range(4))
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    lib.bn_finalize(sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(),
                    shift.data_ptr(), rm.data_ptr(), rv.data_ptr(), C, float(P), 1e-5, 0.1, s)
    zr = z.float().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm2, rv2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    yr = F.batch_norm(zr, rm2, rv2, g_, b_, True, 0.1, 1e-5)
    assert torch.allclose(rm, rm2, atol=1e-4) and torch.allclose(rv, rv2, rtol=1e-3, atol=1e-4)
    res = rnd(P, C, seed=4).half()                  # the skip connection: a forward tensor
    out = torch.empty_like(z)
    lib.bn_apply(z.data_ptr(), scale.data_ptr(), shift.data_ptr(), res.data_ptr(), out.data_ptr(), P, C, 1, s)
    act = F.silu(yr)
    assert relerr(out, act + res.float()) < FWD_TOL
    # pooled mean of the activated output, and backward with SiLU + SE gate terms
    pooled = torch.empty(B, C, device=DEV)
    lib.pool_bn_act(z.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, pooled.data_ptr(), B, HW, C, 1, 1.0 / HW, s)
    assert relerr(pooled, act.view(B, HW, C).mean(1)) < 2e-3
    gate = torch.sigmoid(rnd(B, C, seed=5)).requires_grad_(True)
    dy = rnd(P, C, seed=6).bfloat16()
    sq = act.view(B, HW, C).mean(1)
    extra = rnd(B, C, seed=7)                       # stands for dLoss/d(squeeze) coming back through the SE MLP
    total = ((act.view(B, HW, C) * gate.unsqueeze(1)).reshape(P, C) * dy.float()).sum() + (sq * extra).sum()
    total.backward()
    bs = torch.zeros(2 * C, device=DEV)
    dz = torch.empty(P, C, dtype=torch.bfloat16, device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    lib.bn_bwd(dy.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
               gate.data_ptr(), extra.data_ptr(), HW, 1, bs.data_ptr(), 0, dz.data_ptr(), dg.data_ptr(), db.data_ptr(), P, C, *scr(), s)
    assert relerr(dz, zr.grad) < 1.5e-2
    assert relerr(dg, g_.grad) < 5e-3 and relerr(db, b_.grad) < 5e-3
    dgate = torch.empty(B, C, device=DEV)
    lib.pool_bn_act(z.data_ptr(), scale.data_ptr(), shift.data_ptr(), dy.data_ptr(), dgate.data_ptr(), B, HW, C, 1, 1.0, s)
    assert relerr(dgate, gate.grad) < 5e-3


def test_squeeze_excite_mlp():
    lib, s = _lib()
    B, C, RD = 11, 144, 20
    sq = rnd(B, C, seed=1).requires_grad_(True)
    Wr, br = rnd(RD, C, seed=2, scale=0.2).requires_grad_(True), rnd(RD, seed=3, scale=0.2).requires_grad_(True)
    We, be = rnd(C, RD, seed=4, scale=0.2).requires_grad_(True), rnd(C, seed=5, scale=0.2).requires_grad_(True)
    hr, gate, weT = torch.empty(B, RD, device=DEV), torch.empty(B, C, device=DEV), torch.empty(RD, C, device=DEV)
    hs = torch.empty(B, RD, device=DEV)
    lib.se_mlp_fwd(sq.data_ptr(), Wr.data_ptr(), br.data_ptr(), We.data_ptr(), be.data_ptr(), weT.data_ptr(), hr.data_ptr(),
                   hs.data_ptr(), gate.data_ptr(), B, C, RD, s)
    ref = torch.sigmoid(F.linear(F.silu(F.linear(sq, Wr, br)), We, be))
    assert torch.allclose(gate, ref, atol=1e-5)
    dgate = rnd(B, C, seed=6)
    ref.backward(dgate)
    dr, ds, dweT = torch.empty(B, RD, device=DEV), torch.empty(B, C, device=DEV), torch.empty(RD, C, device=DEV)
    gs = [torch.zeros_like(t) for t in (Wr, br, We, be)]
    lib.se_mlp_bwd(dgate.data_ptr(), gate.data_ptr(), hr.data_ptr(), hs.data_ptr(), sq.data_ptr(), Wr.data_ptr(), weT.data_ptr(), dr.data_ptr(),
                   ds.data_ptr(), dweT.data_ptr(), gs[0].data_ptr(), gs[1].data_ptr(), gs[2].data_ptr(), gs[3].data_ptr(), B, C, RD, s)
    assert torch.allclose(ds, sq.grad, rtol=1e-4, atol=1e-6)
    for g, t in zip(gs, (Wr, br, We, be)):
        assert torch.allclose(g, t.grad, rtol=1e-4, atol=1e-6)


def test_stem_conv_and_transformed_pointwise_gemm():
    lib, s = _lib()
    from multimodalsimilar_amd import ops
    B, H, W, Co = 3, 20, 24, 48
    x = rnd(B, 3, H, W, seed=1)
    w = rnd(Co, 3, 3, 3, seed=2, scale=0.3)
    xr, wr = x.clone(), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=2, padding=1)
    Ho, Wo = ref.shape[2:]
    z = torch.empty(B * Ho * Wo, Co, dtype=torch.float16, device=DEV)
    sums = torch.zeros(2 * Co, device=DEV)
    lib.stem_fwd(x.data_ptr(), w.data_ptr(), z.data_ptr(), sums.data_ptr(), B, H, W, Co, *scr(), s)
    assert relerr(nchw(z, B, Ho, Wo), ref) < FWD_TOL
    assert relerr(sums[:Co], z.float().sum(0)) < 1e-3
    dz = nhwc_g(rnd(B, Co, Ho, Wo, seed=3))
    ref.backward(nchw(dz, B, Ho, Wo))
    dw = torch.zeros_like(w)
    lib.stem_wgrad(dz.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, W, Co, *scr(), s)
    assert relerr(dw, wr.grad) < 2e-3
    # 1x1 conv on silu(scale*z+shift)*gate, forward and weight gradient
    P, Cin, Cout, HW = 360, 144, 40, 30
    z2 = rnd(P, Cin, seed=4).half()
    scale, shift = 1 + 0.1 * rnd(Cin, seed=5), 0.1 * rnd(Cin, seed=6)
    gate = torch.sigmoid(rnd(P // HW, Cin, seed=7))
    w3 = rnd(Cout, Cin, seed=8, scale=0.2).half()
    a = (F.silu(z2.float() * scale + shift).view(P // HW, HW, Cin) * gate.unsqueeze(1)).reshape(P, Cin)
    out = torch.empty(P, Cout, dtype=torch.float16, device=DEV)
    lib.gemm_bf16_xf(1, P, Cout, Cin, z2.data_ptr(), Cin, w3.data_ptr(), Cin, out.data_ptr(), Cout, 0, scale.data_ptr(),
                     shift.data_ptr(), gate.data_ptr(), HW, 1, 0, s)
    assert relerr(out, a @ w3.float().t()) < FWD_TOL
    dz3 = rnd(P, Cout, seed=9).bfloat16()
    gw = torch.zeros(Cout, Cin, device=DEV)
    lib.gemm_bf16_xf(2, Cout, Cin, P, dz3.data_ptr(), Cout, z2.data_ptr(), Cin, gw.data_ptr(), Cin, 1, scale.data_ptr(),
                     shift.data_ptr(), gate.data_ptr(), HW, 2, 1, s)
    assert relerr(gw, dz3.float().t() @ a) < 1.5e-2


def _state_to_oracle(model_sd, prefix="backbone."):
    return {k: v.detach().cpu().float().clone() if v.is_floating_point() else v.detach().cpu().clone()
            for k, v in model_sd.items()}


def l2err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


@pytest.mark.parametrize("name,idx", [("efficientnet_b0", 0), ("efficientnet_b0", 1), ("efficientnet_b0", 2), ("efficientnet_b0", 4),
                                       ("efficientnet_b0", 8), ("efficientnet_b4", 1), ("efficientnet_b4", 10), ("efficientnet_b4", 23),
                                       ("efficientnet_b4", 31)])
def test_mbconv_block_teacher_forced(name, idx):
    """One MBConv block at a time, fed the SAME input as the oracle block (no error carried between blocks):
    output, input gradient and every parameter gradient against torch autograd of oracle.mbconv_forward."""
    from types import SimpleNamespace
    from oracle import effnet_ref
    from multimodalsimilar_amd.effnet import EfficientNet
    model = EfficientNet(name, seed=idx)
    g = torch.Generator().manual_seed(100 + idx)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if p.dim() == 1 and ".bn" in k:
                p.copy_((1.0 if k.endswith("weight") else 0.0) + 0.2 * torch.randn(p.shape, generator=g))
            if ".se." in k and p.dim() == 1:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
    sd = {"backbone." + k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    b = model.arch.blocks[idx]
    ob = effnet_ref.arch(name)["blocks"][idx]
    B, H = 8, 16 if b.stride == 1 else 16
    x = torch.randn(B, b.cin, H, H, generator=g)
    xb = nhwc(x.to(DEV))
    prefix = "backbone." + b.name
    keys = [k for k in sd if k.startswith(prefix + ".") and sd[k].is_floating_point() and "running" not in k]
    sdr = dict(sd)
    for k in keys:
        sdr[k] = sd[k].clone().requires_grad_(True)
    xr = nchw(xb, B, H, H).cpu().requires_grad_(True)
    ref = effnet_ref.mbconv_forward(sdr, prefix, ob, xr, training=True)
    Ho = ref.shape[2]
    dout = nhwc_g(torch.randn(B, b.cout, Ho, Ho, generator=g).to(DEV))
    ref.backward(nchw(dout, B, Ho, Ho).cpu())
    # HIP block
    model._flat.sync_shadow()
    model._bind_grads()
    model._flat.grad.zero_()
    st = SimpleNamespace(B=B, blocks=[])
    st.bnstat = model._buf("bnstat", (4, model._bn_total), torch.float32)
    st.sums_f = model._buf("sums_f", (2 * model._bn_total,), torch.float32); st.sums_f.zero_()
    st.sums_b = model._buf("sums_b", (2 * model._bn_total,), torch.float32); st.sums_b.zero_()
    out, Ho2, Wo2 = model._block_fwd(st, b, xb, H, H)
    assert Ho2 == Ho
    tag = f"mbconv_block_teacher_forced[{name}-{idx}]"
    # forward: fp16 storage of z1 / z2 / (a2) / z3 / the output -- a tenth of north_star's 1e-2 (measured 4-7e-4; bf16 storage: 2-4e-2)
    check(tag, "block output, max-norm relative", relerr(nchw(out, B, Ho, Ho), ref), 2e-3)
    check(tag, "block output, relative L2", l2err(nchw(out, B, Ho, Ho), ref), 1e-3)
    dx = model._block_bwd(st, b, st.blocks[-1], dout)
    check(tag, "input gradient, relative L2 (bf16 gradients)", l2err(nchw(dx, B, H, H), xr.grad), 1e-2)
    named = dict(model.named_parameters())
    gmax = max(sdr[k].grad.norm().item() for k in keys)
    checked, worst = 0, 0.0
    for k in keys:
        gref = sdr[k].grad
        if gref.norm().item() < 1e-5 * gmax:
            continue      # analytically-zero gradient (a constant in front of a BatchNorm)
        e = l2err(named[k[len("backbone."):]].grad, gref)
        worst = max(worst, e)
        assert e < 2e-2, (k, e)
        checked += 1
    record(tag, "worst parameter gradient, relative L2 (bf16 gradients)", worst, 2e-2)
    assert checked >= 8


@pytest.mark.parametrize("name,use_fc", [("efficientnet_b0", True), ("efficientnet_b4", False)])
def test_cv_classifier_matches_oracle(name, use_fc):
    """Whole image tower + top + ArcFace(m=0.2), forward and backward, against the fp32 oracle.

    Per-block parity is established by test_mbconv_block_teacher_forced.  End to end, ~50-100 train-mode BatchNorm
    layers at random init amplify ANY rounding difference: the oracle itself moves by d0 (5-20 % relative L2 on the
    embedding) when bf16 storage is emulated in it (oracle/effnet_ref.py emulate_bf16).  The HIP path must stay
    within that self-measured envelope of the fp32 oracle, and the loss (which sees only angles) within 1e-2."""
    import warnings
    from oracle import effnet_ref, arcface_ref
    from cv_classifier import CvClassifier
    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    model = CvClassifier(name, 64, 50, pretrained=False, use_fc=use_fc)
    g = torch.Generator().manual_seed(1)
    if use_fc:
        model.dropout.p = 0.0
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    B, R = 16, 64
    x = torch.randn(B, 3, R, R, generator=g)
    y = torch.randint(0, 50, (B,), generator=g)
    res = {}
    for emu in (False, True):
        sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
        emb_ref = effnet_ref.cv_predict_emb(sdr, name, x, use_fc=use_fc, training=True, emulate="fp16" if emu else None)
        loss_ref = arcface_ref.ce_loss(arcface_ref.arcface_forward(emb_ref, sdr["classifier.weight"], y, 64.0, 0.2), y)
        loss_ref.backward()
        res[emu] = (emb_ref.detach(), loss_ref.item(), {k: v.grad for k, v in sdr.items() if torch.is_tensor(v) and v.grad is not None})
    emb = model.predict_emb(x.to(DEV))
    loss, _ = model.forward_loss(x.to(DEV), y.to(DEV))
    loss.backward()
    named = dict(model.named_parameters())
    emb_ref, loss_ref, grads = res[False]
    emb_emu, loss_emu, grads_emu = res[True]
    d0 = l2err(emb_emu, emb_ref)
    gmax = max(v.norm().item() for v in grads.values())
    keys = [k for k in named if k in grads and named[k].grad is not None and grads[k].norm().item() > 1e-5 * gmax]
    g0 = sorted(l2err(grads_emu[k], grads[k]) for k in keys)
    ge = sorted(l2err(named[k].grad, grads[k]) for k in keys)
    e = l2err(emb, emb_ref)
    print(f"\n[{name}] emb L2 err {e:.4f} (fp16-storage emulation of the oracle d0 {d0:.4f}); loss {loss.item():.4f} vs {loss_ref:.4f}; "
          f"grad L2 median {ge[len(ge) // 2]:.3f} (envelope {g0[len(g0) // 2]:.3f}) over {len(keys)} tensors")
    assert len(keys) > 50
    # RANDOM-INIT smoke (ill-conditioned on purpose: ~50-100 train-mode BatchNorms at init amplify every rounding).  The tight
    # whole-tower checks are in tests/test_gpu_eval_parity.py (conditioned weights, eval and train mode: 1e-2).  Here: the embedding
    # within twice what fp16 STORAGE alone does to the oracle (+1 %), the loss within north_star's 1e-2, the median parameter
    # gradient (bf16 gradient tensors through ~100 layers) within 8 %.
    tag = f"cv_classifier_matches_oracle[{name}]"
    record(tag, "oracle under fp16-storage emulation: embedding relative L2 (d0)", d0, 1.0)
    check(tag, "embedding relative L2 (random init, train mode)", e, 2.0 * d0 + 0.01)
    check(tag, "loss relative error", abs(loss.item() - loss_ref) / loss_ref, 1e-2)
    check(tag, "median parameter-gradient relative L2", ge[len(ge) // 2], 8e-2)
    # running statistics follow torch semantics (momentum 0.1, unbiased variance); two forward passes were run
    stats = {}
    effnet_ref.cv_predict_emb(sd, name, x, use_fc=use_fc, training=True, stats=stats)
    mu, var = stats["backbone.bn1"]
    n = B * (R // 2) ** 2
    rm, rv = model.backbone.bn1.running_mean.cpu(), model.backbone.bn1.running_var.cpu()
    assert torch.allclose(rm, 0.1 * mu * 1.9, atol=2e-3)
    assert torch.allclose(rv, 0.81 + 0.19 * var * n / (n - 1), rtol=2e-2, atol=2e-3)
    assert int(model.backbone.bn1.num_batches_tracked) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("P,Cin,Cout,xf", [(1000, 24, 144, 0), (6272, 96, 40, 1), (513, 48, 24, 0), (12544, 272, 448, 1)])
def test_pointwise_conv_with_bn_statistics(P, Cin, Cout, xf):
    """conv_pw / conv_pwl + the statistics pass of the following BatchNorm2d in one launch: output equals the plain GEMM,
    sums equal sum / sum of squares of the fp16-rounded output (ragged M, N tiles included)."""
    torch.manual_seed(P + Cin)
    hw = 49
    B = (P + hw - 1) // hw
    x = torch.randn(P, Cin, device=DEV).half()
    w = (torch.randn(Cout, Cin, device=DEV) * 0.2).half()
    sc, sh = torch.rand(Cin, device=DEV) + 0.5, torch.randn(Cin, device=DEV) * 0.3
    gate = torch.rand(B, Cin, device=DEV)
    z = torch.empty(P, Cout, dtype=torch.float16, device=DEV)
    sums = torch.zeros(2 * Cout, device=DEV)
    lib, s = _lib()
    lib.gemm_bf16_bnstats(xf, P, Cout, Cin, x.data_ptr(), Cin, w.data_ptr(), Cin, z.data_ptr(), Cout,
                          sc.data_ptr() if xf else None, sh.data_ptr() if xf else None, gate.data_ptr() if xf else None, hw,
                          sums.data_ptr(), *scr(), s)
    a = x.float()
    if xf:
        a = torch.nn.functional.silu(a * sc + sh) * gate.repeat_interleave(hw, 0)[:P]
        a = a.half().float()
    ref = a @ w.float().t()
    assert relerr(z, ref) < FWD_TOL
    zf = z.float()
    ssum, ssq = zf.double().sum(0), (zf.double() ** 2).sum(0)
    assert (sums[:Cout].double() - ssum).abs().max() < 1e-4 * zf.abs().double().sum(0).max() + 1e-3      # fp32 summation order
    assert (sums[Cout:].double() - ssq).abs().max() < 1e-4 * ssq.max() + 1e-3


def test_squeeze_that_keeps_the_activation_and_gate_only_operand():
    """mmsim_pool_bn_act_store (SE squeeze + a2 = silu(bn(z2)) kept) and the gate-only operand transform of the projection
    conv (forward and weight gradient) against the recompute form and fp32 torch."""
    lib, s = _lib()
    B, HW, Cin, Cout = 6, 49, 144, 40
    P = B * HW
    z2 = rnd(P, Cin, seed=4).half()
    scale, shift = 1 + 0.1 * rnd(Cin, seed=5), 0.1 * rnd(Cin, seed=6)
    a_ref = F.silu(z2.float() * scale + shift)
    a2 = torch.empty(P, Cin, dtype=torch.float16, device=DEV)
    sq = torch.empty(B, Cin, device=DEV)
    lib.pool_bn_act_store(z2.data_ptr(), scale.data_ptr(), shift.data_ptr(), a2.data_ptr(), sq.data_ptr(), B, HW, Cin, 1.0 / HW, s)
    assert relerr(a2, a_ref) < FWD_TOL
    assert relerr(sq, a_ref.view(B, HW, Cin).mean(1)) < 1e-3
    sq2 = torch.empty(B, Cin, device=DEV)
    lib.pool_bn_act(z2.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, sq2.data_ptr(), B, HW, Cin, 1, 1.0 / HW, s)
    assert torch.equal(sq, sq2)                      # the pooled value does not depend on whether a2 is kept
    gate = torch.sigmoid(rnd(B, Cin, seed=7))
    w3 = rnd(Cout, Cin, seed=8, scale=0.2).half()
    ag = (a2.float().view(B, HW, Cin) * gate.unsqueeze(1)).reshape(P, Cin)
    out = torch.empty(P, Cout, dtype=torch.float16, device=DEV)
    lib.gemm_bf16_xf(1, P, Cout, Cin, a2.data_ptr(), Cin, w3.data_ptr(), Cin, out.data_ptr(), Cout, 0, None, None, gate.data_ptr(),
                     HW, 1, 0, s)
    assert relerr(out, ag @ w3.float().t()) < FWD_TOL
    dz3 = rnd(P, Cout, seed=9).bfloat16()
    gw = torch.zeros(Cout, Cin, device=DEV)
    lib.gemm_bf16_xf(2, Cout, Cin, P, dz3.data_ptr(), Cout, a2.data_ptr(), Cin, gw.data_ptr(), Cin, 1, None, None, gate.data_ptr(),
                     HW, 2, 1, s)
    assert relerr(gw, dz3.float().t() @ ag) < 1.5e-2
    from multimodalsimilar_amd._lib import MmsimError
    with pytest.raises(MmsimError):                  # neither scale/shift nor a gate: rejected before launch
        lib.gemm_bf16_xf(1, P, Cout, Cin, a2.data_ptr(), Cin, w3.data_ptr(), Cin, out.data_ptr(), Cout, 0, None, None, None, HW, 1, 0, s)


@pytest.mark.parametrize("P,Cin,Cout,HW,xf", [(1568, 200, 72, 49, 2), (1568, 200, 72, 49, 0), (392, 1632, 272, 49, 2), (6272, 112, 672, 196, 0)])
def test_paired_backward_products_equal_the_separate_launches(P, Cin, Cout, HW, xf):
    """mmsim_gemm_group_begin/_end: the (weight gradient, data gradient) pair of a 1x1 conv as one launch gives exactly the
    results of the two launches (split-K 1, so no atomic-order noise), with and without the gated operand transform."""
    from multimodalsimilar_amd import ops
    lib, s = _lib()
    dy = rnd(P, Cout, seed=1).bfloat16()
    xin = rnd(P, Cin, seed=2).half()                # the activation operand of the weight gradient: fp16
    gate = torch.sigmoid(rnd(P // HW, Cin, seed=3))
    w = rnd(Cout, Cin, seed=4, scale=0.2).bfloat16()
    res = torch.randn(P, Cin, device=DEV).bfloat16()

    def run(grouped):
        gw = torch.zeros(Cout, Cin, device=DEV)
        dx = torch.empty(P, Cin, dtype=torch.bfloat16, device=DEV)
        if grouped:
            lib.gemm_group_begin()
        if xf == 2:
            lib.gemm_bf16_xf(2, Cout, Cin, P, dy.data_ptr(), Cout, xin.data_ptr(), Cin, gw.data_ptr(), Cin, 1, None, None,
                             gate.data_ptr(), HW, 1, 1, s)
        else:
            ops.gemm(dy, xin, gw, trans_a=True, b_kmajor=False, split_k=1, accumulate=True)
        ops.gemm(dy, w, dx, b_kmajor=False, epilogue=ops.EPI_ADD, aux_in=res)
        if grouped:
            lib.gemm_group_end()
        torch.cuda.synchronize()
        return gw, dx

    gw0, dx0 = run(False)
    gw1, dx1 = run(True)
    assert torch.equal(gw0, gw1) and torch.equal(dx0, dx1)
    a = xin.float() if xf == 0 else (xin.float().view(P // HW, HW, Cin) * gate.unsqueeze(1)).reshape(P, Cin)
    assert relerr(gw1, dy.float().t() @ a) < 1.5e-2
    assert relerr(dx1, dy.float() @ w.float() + res.float()) < 1.5e-2
    # a group that is not a (wgrad, dgrad) pair falls back to program-order launches; an unmatched _end is an error
    lib.gemm_group_begin()
    ops.gemm(dy, w, dx1, b_kmajor=False)
    lib.gemm_group_end()
    torch.cuda.synchronize()
    assert relerr(dx1, dy.float() @ w.float()) < 1.5e-2
    with pytest.raises(RuntimeError):
        lib.gemm_group_end()


@pytest.mark.parametrize("B,HW,mid,cout", [(3, 256, 48, 24), (2, 512, 24, 24), (5, 64, 192, 32), (4, 128, 144, 32), (6, 32, 192, 56),
                                           (3, 96, 192, 56), (2, 784, 176, 40)])
def test_streaming_projection_forward_equals_the_gemm_form(B, HW, mid, cout):
    """mmsim_pw_project_fwd (early-stage projection conv as one streaming pass: gate applied while staging, W3 resident in LDS,
    BatchNorm statistics in registers) against fp32 torch and against mmsim_gemm_bf16_bnstats, the form it replaces: same
    bf16 outputs up to the accumulation order, same statistics.  Shapes cover the three tile variants, strips that straddle two
    images (HW not a multiple of the strip) and a partly empty channel tile (cout = 24, 56)."""
    lib, s = _lib()
    P = B * HW
    assert lib.pw_project_fwd_eligible(P, HW, mid, cout)
    a2 = rnd(P, mid, seed=1).half()
    gate = torch.sigmoid(rnd(B, mid, seed=2))
    w3 = rnd(cout, mid, seed=3, scale=0.2).half()
    z3 = torch.empty(P, cout, dtype=torch.float16, device=DEV)
    sums = torch.zeros(2 * cout, device=DEV)
    lib.pw_project_fwd(a2.data_ptr(), gate.data_ptr(), w3.data_ptr(), z3.data_ptr(), sums.data_ptr(), P, HW, mid, cout, *scr(), s)
    a = ((a2.float().view(B, HW, mid) * gate.unsqueeze(1)).half().float()).reshape(P, mid)      # the operand as staged (fp16)
    ref = a @ w3.float().t()
    assert relerr(z3, ref) < FWD_TOL
    assert relerr(sums[:cout], z3.float().sum(0)) < 1e-4 and relerr(sums[cout:], (z3.float() ** 2).sum(0)) < 1e-4
    z3g = torch.empty_like(z3)
    sums_g = torch.zeros_like(sums)
    lib.gemm_bf16_bnstats(1, P, cout, mid, a2.data_ptr(), mid, w3.data_ptr(), mid, z3g.data_ptr(), cout, None, None, gate.data_ptr(), HW,
                          sums_g.data_ptr(), *scr(), s)
    assert relerr(z3, z3g) < 1e-3 and relerr(sums, sums_g) < 2e-3
    assert not lib.pw_project_fwd_eligible(P, HW, 1632, 272) and not lib.pw_project_fwd_eligible(P, HW, mid, cout + 4)


@pytest.mark.parametrize("B,HW,mid,cout", [(3, 256, 48, 24), (2, 384, 24, 24), (5, 64, 192, 32), (4, 128, 144, 32), (4, 96, 192, 24)])
def test_streaming_projection_backward_equals_the_gemm_form(B, HW, mid, cout):
    """mmsim_pw_project_bwd: d(a2*gate) = dz3 W3 and dW3 += dz3^T (a2*gate) out of one streaming pass, against fp32 torch and
    against the two generic products it replaces (split-K 1).  Covers both tile variants, strips straddling two images and a
    partly empty output-channel tile; dW3 is accumulated onto a non-zero buffer."""
    from multimodalsimilar_amd import ops
    lib, s = _lib()
    P = B * HW
    assert lib.pw_project_bwd_eligible(P, HW, mid, cout)
    a2 = rnd(P, mid, seed=1).half()
    gate = torch.sigmoid(rnd(B, mid, seed=2))
    w3 = rnd(cout, mid, seed=3, scale=0.2).bfloat16()      # the data gradient reads the bf16 weight shadow
    dz3 = rnd(P, cout, seed=4).bfloat16()
    base = rnd(cout, mid, seed=5)
    da = torch.empty(P, mid, dtype=torch.bfloat16, device=DEV)
    dw = base.clone()
    lib.pw_project_bwd(dz3.data_ptr(), a2.data_ptr(), gate.data_ptr(), w3.data_ptr(), da.data_ptr(), dw.data_ptr(), P, HW, mid, cout, *scr(), s)
    ag = ((a2.float().view(B, HW, mid) * gate.unsqueeze(1)).bfloat16().float()).reshape(P, mid)
    assert relerr(da, dz3.float() @ w3.float()) < 1e-2
    assert relerr(dw - base, dz3.float().t() @ ag) < 5e-3
    da_g = torch.empty_like(da)
    dw_g = base.clone()
    lib.gemm_bf16_xf(2, cout, mid, P, dz3.data_ptr(), cout, a2.data_ptr(), mid, dw_g.data_ptr(), mid, 1, None, None, gate.data_ptr(), HW, 1, 1, s)
    ops.gemm(dz3, w3, da_g, b_kmajor=False)
    assert relerr(da, da_g) < 4e-3 and relerr(dw - base, dw_g - base) < 2e-3
    assert not lib.pw_project_bwd_eligible(P, HW, 336, 56)


@pytest.mark.parametrize("P,mid,cin", [(640, 192, 32), (1024, 144, 24), (64, 64, 8), (4096, 176, 16)])
def test_streaming_expansion_forward_equals_the_gemm_form(P, mid, cin):
    """mmsim_pw_expand_fwd (56^2-stage expansion conv as an output stream: W1 in LDS, output tile assembled in LDS, statistics kept by
    the threads that copy it out) against fp32 torch and mmsim_gemm_bf16_bnstats."""
    lib, s = _lib()
    assert lib.pw_expand_fwd_eligible(P, mid, cin)
    x = rnd(P, cin, seed=1).half()
    w1 = rnd(mid, cin, seed=2, scale=0.3).half()
    z1 = torch.empty(P, mid, dtype=torch.float16, device=DEV)
    sums = torch.zeros(2 * mid, device=DEV)
    lib.pw_expand_fwd(x.data_ptr(), w1.data_ptr(), z1.data_ptr(), sums.data_ptr(), P, mid, cin, *scr(), s)
    assert relerr(z1, x.float() @ w1.float().t()) < FWD_TOL
    assert relerr(sums[:mid], z1.float().sum(0)) < 1e-4 and relerr(sums[mid:], (z1.float() ** 2).sum(0)) < 1e-4
    z1g = torch.empty_like(z1)
    sums_g = torch.zeros_like(sums)
    lib.gemm_bf16_bnstats(0, P, mid, cin, x.data_ptr(), cin, w1.data_ptr(), cin, z1g.data_ptr(), mid, None, None, None, 1,
                          sums_g.data_ptr(), *scr(), s)
    assert torch.equal(z1, z1g) or relerr(z1, z1g) < 1e-3                 # one 32-deep MFMA step either way: usually bit-equal
    assert relerr(sums, sums_g) < 1e-3
    assert not lib.pw_expand_fwd_eligible(P, 336, 56) and not lib.pw_expand_fwd_eligible(P + 8, mid, cin)


@pytest.mark.parametrize("B,HW,mid,cout", [(3, 256, 48, 24), (5, 64, 192, 32), (4, 96, 144, 24)])
def test_streaming_projection_with_the_activation_formed_on_the_fly(B, HW, mid, cout):
    """mmsim_pw_project_fwd_xf / _bwd_xf: the operand silu(scale z2 + shift) * gate is formed while the strip is staged (a2 never
    stored).  Against fp32 torch and against the stored-a2 kernels fed with a2 = bf16(silu(scale z2 + shift))."""
    lib, s = _lib()
    P = B * HW
    z2 = rnd(P, mid, seed=1).half()
    scale, shift = 1 + 0.1 * rnd(mid, seed=6), 0.1 * rnd(mid, seed=7)
    gate = torch.sigmoid(rnd(B, mid, seed=2))
    w3 = rnd(cout, mid, seed=3, scale=0.2).half()
    w3b = w3.float().bfloat16()                            # what the bf16 shadow of the same weights holds (backward)
    act = F.silu(z2.float() * scale + shift)
    ag = (act.view(B, HW, mid) * gate.unsqueeze(1)).reshape(P, mid)
    z3 = torch.empty(P, cout, dtype=torch.float16, device=DEV)
    sums = torch.zeros(2 * cout, device=DEV)
    lib.pw_project_fwd_xf(z2.data_ptr(), scale.data_ptr(), shift.data_ptr(), gate.data_ptr(), w3.data_ptr(), z3.data_ptr(),
                          sums.data_ptr(), P, HW, mid, cout, *scr(), s)
    assert relerr(z3, ag @ w3.float().t()) < FWD_TOL
    assert relerr(sums[:cout], z3.float().sum(0)) < 1e-4
    a2 = act.half()
    z3s = torch.empty_like(z3)
    sums_s = torch.zeros_like(sums)
    lib.pw_project_fwd(a2.data_ptr(), gate.data_ptr(), w3.data_ptr(), z3s.data_ptr(), sums_s.data_ptr(), P, HW, mid, cout, *scr(), s)
    assert relerr(z3, z3s) < 1e-3                      # one rounding (fp32 activation x gate) instead of two
    dz3 = rnd(P, cout, seed=4).bfloat16()
    da = torch.empty(P, mid, dtype=torch.bfloat16, device=DEV)
    dw = torch.zeros(cout, mid, device=DEV)
    lib.pw_project_bwd_xf(dz3.data_ptr(), z2.data_ptr(), scale.data_ptr(), shift.data_ptr(), gate.data_ptr(), w3b.data_ptr(),
                          da.data_ptr(), dw.data_ptr(), P, HW, mid, cout, *scr(), s)
    assert relerr(da, dz3.float() @ w3b.float()) < 1e-2
    assert relerr(dw, dz3.float().t() @ ag) < 5e-3


@pytest.mark.parametrize("M,N,K", [(300, 72, 200), (256, 512, 1792), (1000, 40, 24)])
def test_fp16_forward_product_and_mixed_weight_gradient(M, N, K):
    """mmsim_gemm_fmt: fmt 1 (fp16 x fp16 -> fp16 / f32 + bias: the forward 1x1 convs and the fc layer of cv_classifier.py:53) and
    fmt 2 (bf16 gradient^T x fp16 activation -> f32: their weight gradients) against fp32 torch; wrong layouts are refused."""
    from multimodalsimilar_amd import ops
    from multimodalsimilar_amd._lib import MmsimError
    a = rnd(M, K, seed=1).half()
    b = rnd(N, K, seed=2, scale=0.2).half()
    ref = a.float() @ b.float().t()
    c16 = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
    ops.gemm(a, b, c16)
    assert relerr(c16, ref) < FWD_TOL
    bias = rnd(N, seed=3)
    c32 = torch.empty(M, N, device=DEV)
    ops.gemm(a, b, c32, bias=bias)
    assert relerr(c32, ref + bias) < 2e-4                       # fp32 accumulation of exact fp16 products
    dy = rnd(M, N, seed=4).bfloat16()
    gw = torch.full((N, K), 0.25, device=DEV)
    ops.gemm(dy, a, gw, trans_a=True, b_kmajor=False, split_k=2, accumulate=True)
    assert relerr(gw - 0.25, dy.float().t() @ a.float().bfloat16().float()) < 2e-3      # the activation enters the MFMA as bf16
    with pytest.raises(ValueError):
        ops.gemm(a, b, c16, b_kmajor=False)                     # fp16 operands: forward layout only
    with pytest.raises(TypeError):
        ops.gemm(a, b, torch.empty(M, N, dtype=torch.bfloat16, device=DEV))
    lib, s = _lib()
    with pytest.raises(MmsimError):                             # the C ABI refuses it too
        lib.gemm_fmt(1, 1, 0, M, N, K, a.data_ptr(), K, b.data_ptr(), K, c32.data_ptr(), N, 1, None, 0, None, None, 0, 1.0, 1, 0, s)


def test_adamw_writes_both_weight_shadows():
    """mmsim_adamw_step2: the bf16 and the fp16 copies of the updated parameters (image tower: forward reads fp16, dgrad bf16)."""
    from multimodalsimilar_amd import ops
    n = 4096 + 8
    p, g = rnd(n, seed=1), rnd(n, seed=2, scale=0.1)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sb, sh = torch.empty(n, dtype=torch.bfloat16, device=DEV), torch.empty(n, dtype=torch.float16, device=DEV)
    ops.adamw_step(p, g, m, v, sb, 1e-2, 0.9, 0.999, 1e-8, 0.01, 1, shadow16=sh)
    assert torch.equal(sb, p.bfloat16()) and torch.equal(sh, p.half())
    big = torch.full((8,), 1e6, device=DEV)
    out = torch.empty(8, dtype=torch.float16, device=DEV)
    ops.cast_to_f16(big, out)
    assert torch.isfinite(out).all() and float(out[0]) == 65504.0      # saturating, never inf
