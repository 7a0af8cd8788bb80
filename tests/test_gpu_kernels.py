"""Per-kernel parity on the MI355X: each HIP kernel (called through the C ABI) against a plain fp32 PyTorch
reference of the same op on the same (bf16-rounded) inputs.  Tolerances: bf16 outputs 1e-2 of the output scale,
fp32 outputs 2e-3 (bf16 products, fp32 accumulation) -- BASELINE.json north_star: 1e-3 fp32 / 1e-2 bf16."""
import math
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from multimodalsimilar_amd import ops
    return ops


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def rnd(*s, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (200, 136, 72), (1000, 40, 128), (8, 1000, 768),
                                   (513, 257, 1032),
                                   # tile-aligned shapes take the LDS-DMA fast path (256x128x64 tiles)
                                   (256, 128, 64), (512, 384, 192), (1024, 256, 1024), (768, 1280, 128),
                                   # ... and with >= 128 tiles of 256x256 the 256x256x32 kernel (1, 2, 3, 5 K-slices: ring edge cases)
                                   (8192, 1024, 32), (8192, 1024, 64), (8192, 1024, 96), (4096, 2048, 160)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_layouts(M, N, K, layout):
    ops = _ops()
    if layout == "tn" and M % 8:
        M = (M + 7) // 8 * 8
    if layout != "nt" and N % 8:
        N = (N + 7) // 8 * 8
    a = rnd(M, K, seed=1).bfloat16()
    b = rnd(N, K, seed=2).bfloat16()
    ref = a.float() @ b.float().t()
    if layout == "nt":
        A, Bm, kw = a, b, dict()
    elif layout == "nn":
        A, Bm, kw = a, b.t().contiguous(), dict(b_kmajor=False)
    else:
        A, Bm, kw = a.t().contiguous(), b.t().contiguous(), dict(trans_a=True, b_kmajor=False)
    c = ops.alloc_2d(M, N, torch.float32, DEV)
    ops.gemm(A, Bm, c, **kw)
    assert relerr(c, ref) < 2e-3
    cb = ops.alloc_2d(M, N, torch.bfloat16, DEV)
    ops.gemm(A, Bm, cb, **kw)
    assert relerr(cb, ref) < 1e-2


def test_gemm_epilogues_and_splitk():
    ops = _ops()
    M, N, K = 384, 256, 320
    a = rnd(M, K, seed=3).bfloat16(); b = rnd(N, K, seed=4, scale=0.1).bfloat16()
    bias = rnd(N, seed=5)
    pre = a.float() @ b.float().t() + bias
    # bias + GELU with pre-activation side output
    c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV); aux = torch.empty_like(c)
    ops.gemm(a, b, c, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux)
    assert relerr(aux, pre) < 1e-2 and relerr(c, F.gelu(pre)) < 1e-2
    # tanh, f32 out
    cf = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(a, b, cf, bias=bias, epilogue=ops.EPI_TANH)
    assert (cf - torch.tanh(pre)).abs().max() < 5e-3
    # multiply by gelu'(aux_in)
    x = aux.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    ops.gemm(a, b, c, epilogue=ops.EPI_MUL_GELU_GRAD, aux_in=aux)
    assert relerr(c, (pre - bias) * x.grad) < 1e-2
    # residual add
    r = rnd(M, N, seed=6).bfloat16()
    ops.gemm(a, b, c, epilogue=ops.EPI_ADD, aux_in=r)
    assert relerr(c, pre - bias + r.float()) < 1e-2
    # the pair the text tower trains with: GELU that saves gelu'(pre) (of the ROUNDED pre-activation), and the plain multiply
    dg = torch.empty_like(c)
    ops.gemm(a, b, c, bias=bias, epilogue=ops.EPI_GELU_DGELU, aux_out=dg)
    xr = pre.bfloat16().float().requires_grad_(True)
    F.gelu(xr).sum().backward()
    assert relerr(c, F.gelu(pre)) < 1e-2 and relerr(dg, xr.grad) < 1e-2
    ops.gemm(a, b, c, epilogue=ops.EPI_MUL, aux_in=dg)
    assert relerr(c, (pre - bias) * dg.float()) < 1e-2
    # split-K atomic accumulation on top of existing contents, and plain accumulate
    at = a.t().contiguous()        # [K, M] -> product over K as the slow index
    bt = b.t().contiguous()
    base = rnd(M, N, seed=7)
    out = base.clone()
    ops.gemm(at, bt, out, trans_a=True, b_kmajor=False, split_k=4, accumulate=True)
    assert relerr(out, base + pre - bias) < 2e-3
    out = base.clone()
    ops.gemm(at, bt, out, trans_a=True, b_kmajor=False, split_k=1, accumulate=True)
    assert relerr(out, base + pre - bias) < 2e-3


@pytest.mark.parametrize("layout", ["nt", "nn"])
@pytest.mark.parametrize("M,N,K,sk", [(200, 136, 1024, 4), (513, 264, 1096, 3), (128, 128, 96, 1), (1000, 40, 224, 2)])
def test_generic_ring_gemm_ragged_edges_and_splitk(M, N, K, sk, layout):
    """The LDS-DMA ring form of the generic 128 x 128 product (gemm_ring_kernel: forward and data-gradient layouts, K >= 96 per split):
    ragged M / N (rows re-read, outputs predicated), K tails of 8 / 16 elements (zero chunks as DMA source), split-K atomics on top of
    existing contents -- against fp32 torch."""
    ops = _ops()
    a = rnd(M, K, seed=11).bfloat16()
    b = rnd(N, K, seed=12, scale=0.1).bfloat16()
    ref = a.float() @ b.float().t()
    Bm, kw = (b, dict()) if layout == "nt" else (b.t().contiguous(), dict(b_kmajor=False))
    base = rnd(M, N, seed=13)
    out = base.clone()
    ops.gemm(a, Bm, out, split_k=sk, accumulate=True, **kw)
    assert relerr(out, base + ref) < 2e-3
    cb = ops.alloc_2d(M, N, torch.bfloat16, DEV)
    r = rnd(M, N, seed=14).bfloat16()
    ops.gemm(a, Bm, cb, epilogue=ops.EPI_ADD, aux_in=r, **kw)
    assert relerr(cb, ref + r.float()) < 1e-2


def test_gemm_rejects_bad_arguments():
    ops = _ops()
    from multimodalsimilar_amd import MmsimError
    a = torch.zeros(16, 12, dtype=torch.bfloat16, device=DEV)     # lda = 12: not a multiple of 8
    b = torch.zeros(16, 12, dtype=torch.bfloat16, device=DEV)
    c = torch.zeros(16, 16, dtype=torch.float32, device=DEV)
    with pytest.raises(MmsimError):
        ops.gemm(a, b, c)
    with pytest.raises(ValueError):
        ops.gemm(torch.zeros(16, 16, dtype=torch.bfloat16, device=DEV), torch.zeros(8, 24, dtype=torch.bfloat16, device=DEV), c)
    with pytest.raises(MmsimError):
        ops.gemm(a.cpu(), b.cpu(), c.cpu())


def _attn_ref(qkv, mask, B, S, nh, H):
    q, k, v = qkv.float().view(B, S, 3, nh, 64).permute(2, 0, 3, 1, 4)
    sc = q @ k.transpose(-1, -2) / 8.0
    if mask is not None:
        sc = sc + torch.zeros(B, 1, 1, S, device=qkv.device).masked_fill(mask.view(B, 1, 1, S) == 0, float("-inf"))
    p = torch.softmax(sc, -1)
    return (p @ v).transpose(1, 2).reshape(B * S, H)


@pytest.mark.parametrize("S", [32, 64, 128])
@pytest.mark.parametrize("masked", [False, True])
def test_attention_fwd_bwd(S, masked):
    ops = _ops()
    B, nh = 3, 2
    H = nh * 64
    qkv = rnd(B * S, 3 * H, seed=S).bfloat16()
    mask = None
    if masked:
        lens = torch.tensor([S, S // 2 + 3, 5])
        mask = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).long().to(DEV)
    ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * nh * S, dtype=torch.float32, device=DEV)
    ops.attn_fwd(qkv, mask, ctx, lse, B, S, nh, H)
    x = qkv.float().requires_grad_(True)
    ref = _attn_ref(x, mask, B, S, nh, H)
    assert relerr(ctx, ref) < 1e-2
    dctx = rnd(B * S, H, seed=S + 1).bfloat16()
    ref.backward(dctx.float())
    dqkv = torch.empty_like(qkv)
    ops.attn_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, nh, H)
    g = x.grad
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        assert relerr(dqkv[:, sl], g[:, sl]) < 2e-2, name
    # fused bias gradient: identical dqkv, dbias += column sums of the rounded values
    dqkv2 = torch.empty_like(qkv)
    dbias = torch.full((3 * H,), 0.5, device=DEV)
    ops.attn_bwd(qkv, mask, ctx, dctx, lse, dqkv2, B, S, nh, H, dbias=dbias)
    assert torch.equal(dqkv2, dqkv)
    want = 0.5 + dqkv.float().double().sum(0)
    assert (dbias.double() - want).abs().max() < 1e-4 * dqkv.float().abs().double().sum(0).max() + 1e-3


def test_attention_dropout_statistics_and_consistency():
    ops = _ops()
    B, nh, S = 2, 2, 128
    H = nh * 64
    qkv = rnd(B * S, 3 * H, seed=9).bfloat16()
    # v = 1 everywhere: ctx = sum_k dropped-P = (#kept mass)/(1-p); its mean over many rows estimates 1
    qkv[:, 2 * H:] = 1.0
    ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * nh * S, dtype=torch.float32, device=DEV)
    ops.attn_fwd(qkv, None, ctx, lse, B, S, nh, H, dropout_p=0.1, seed=123, stream_id=7)
    m = ctx.float().mean().item()
    assert abs(m - 1.0) < 0.02
    ctx2 = torch.empty_like(ctx)
    ops.attn_fwd(qkv, None, ctx2, lse, B, S, nh, H, dropout_p=0.1, seed=123, stream_id=7)
    assert torch.equal(ctx, ctx2)                       # same key -> same mask
    ops.attn_fwd(qkv, None, ctx2, lse, B, S, nh, H, dropout_p=0.1, seed=124, stream_id=7)
    assert not torch.equal(ctx, ctx2)


@pytest.mark.parametrize("H", [128, 256, 768, 1024])
def test_add_ln_fwd_bwd(H):
    ops = _ops()
    M = 77
    t = rnd(M, H, seed=1).bfloat16(); r = rnd(M, H, seed=2).bfloat16()
    gamma = 1 + 0.1 * rnd(H, seed=3); beta = 0.1 * rnd(H, seed=4)
    y = torch.empty_like(t); h = torch.empty_like(t)
    mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
    ops.add_ln_fwd(t, r, gamma, beta, y, h, mean, rstd, 1e-12)
    yr = (t.float() + r.float()).bfloat16().float().requires_grad_(True)
    g_ = gamma.clone().requires_grad_(True); b_ = beta.clone().requires_grad_(True)
    hr = F.layer_norm(yr, (H,), g_, b_, 1e-12)
    assert relerr(h, hr) < 1e-2 and relerr(y, yr) < 1e-2
    da = rnd(M, H, seed=5).bfloat16(); db = rnd(M, H, seed=6).bfloat16()
    hr.backward(da.float() + db.float())
    dy = torch.empty_like(t)
    dg = torch.zeros(H, device=DEV); dbt = torch.zeros(H, device=DEV); dbias = torch.zeros(H, device=DEV)
    ops.ln_bwd(da, db, y, mean, rstd, gamma, dy, None, dg, dbt, dbias)
    assert relerr(dy, yr.grad) < 1e-2
    assert relerr(dg, g_.grad) < 1e-2 and relerr(dbt, b_.grad) < 1e-2
    assert relerr(dbias, dy.float().sum(0)) < 1e-3


def test_hidden_dropout_mask_is_replayed_in_backward():
    ops = _ops()
    M, H, p = 64, 256, 0.25
    t = torch.ones(M, H, dtype=torch.bfloat16, device=DEV); r = torch.zeros_like(t)
    gamma = torch.ones(H, device=DEV); beta = torch.zeros(H, device=DEV)
    y = torch.empty_like(t); h = torch.empty_like(t)
    mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
    ops.add_ln_fwd(t, r, gamma, beta, y, h, mean, rstd, 1e-12, p, 42, 3)
    keep = (y.float() != 0)
    assert abs(keep.float().mean().item() - (1 - p)) < 0.02
    assert torch.allclose(y.float()[keep], torch.full_like(y.float()[keep], 1 / (1 - p)), rtol=1e-2)
    dy = torch.empty_like(t); dt = torch.empty_like(t)
    dh = rnd(M, H, seed=8).bfloat16()
    z = torch.zeros(H, device=DEV)
    ops.ln_bwd(dh, None, y, mean, rstd, gamma, dy, dt, z.clone(), z.clone(), z.clone(), p, 42, 3)
    assert torch.equal(dt.float() != 0, keep & (dy.float() != 0))
    assert relerr(dt.float()[keep], dy.float()[keep] / (1 - p)) < 1e-2


@pytest.mark.parametrize("with_pids", [False, True])
def test_embed_ln_fwd_bwd(with_pids):
    ops = _ops()
    B, S, H, V = 5, 32, 256, 50
    word = rnd(V, H, seed=1, scale=0.5); pos = rnd(64, H, seed=2, scale=0.5); typ = rnd(2, H, seed=3, scale=0.5)
    gamma = 1 + 0.1 * rnd(H, seed=4); beta = 0.1 * rnd(H, seed=5)
    ids = torch.randint(0, V, (B, S), device=DEV); tts = torch.randint(0, 2, (B, S), device=DEV)
    # explicit position ids (nlp_classifier.py:23-27 forwards them): arbitrary rows of the position table, repeated ones too
    pids = torch.randint(0, 64, (B, S), device=DEV) if with_pids else None
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV)
    ops.embed_ln_fwd(ids, tts, word, pos, typ, gamma, beta, out, B, S, H, 1e-12, err, pids=pids)
    ws = [t.clone().requires_grad_(True) for t in (word, pos, typ, gamma, beta)]
    pe = ws[1][pids] if with_pids else ws[1][torch.arange(S, device=DEV)].unsqueeze(0)
    e = ws[0][ids] + ws[2][tts] + pe
    ref = F.layer_norm(e, (H,), ws[3], ws[4], 1e-12).view(B * S, H)
    assert relerr(out, ref) < 1e-2
    dout = rnd(B * S, H, seed=6).bfloat16()
    ref.backward(dout.float())
    gs = [torch.zeros_like(t) for t in (word, pos, typ, gamma, beta)]
    ops.embed_ln_bwd(dout, ids, tts, word, pos, typ, gamma, gs[0], gs[1], gs[2], gs[3], gs[4], B, S, H, 1e-12, err, pids=pids)
    for g, w, n in zip(gs, ws, ("word", "pos", "type", "gamma", "beta")):
        assert relerr(g, w.grad) < 2e-3, n
    assert int(err.item()) == 0


def test_embed_out_of_range_index_raises_flag_and_stays_inside_the_tables():
    """A token id >= vocab (tokenizer / vocab mismatch) must neither read nor scatter-add outside the word table: the kernels
    clamp it and raise the device flag (nn.Embedding raises an IndexError: BertModel.check_indices / check_labels read it)."""
    ops = _ops()
    B, S, H, V = 2, 32, 256, 50
    word = rnd(V, H, seed=1, scale=0.5); pos = rnd(32, H, seed=2, scale=0.5); typ = rnd(2, H, seed=3, scale=0.5)
    gamma = torch.ones(H, device=DEV); beta = torch.zeros(H, device=DEV)
    ids = torch.randint(0, V, (B, S), device=DEV)
    ids[1, 7] = V + 1000
    ids[0, 3] = -5
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV)
    ops.embed_ln_fwd(ids, None, word, pos, typ, gamma, beta, out, B, S, H, 1e-12, err)
    assert int(err.item()) == 1 and torch.isfinite(out.float()).all()
    # guard band after the word-gradient table: the scatter-add of the bad row must not land in it
    buf = torch.zeros(V * H + 4096, device=DEV)
    gs = [buf[:V * H].view(V, H), torch.zeros_like(pos), torch.zeros_like(typ), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)]
    err.zero_()
    ops.embed_ln_bwd(rnd(B * S, H, seed=6).bfloat16(), ids, None, word, pos, typ, gamma, *gs, B, S, H, 1e-12, err)
    assert int(err.item()) == 1 and float(buf[V * H:].abs().max()) == 0.0


def test_bert_forwards_position_ids_and_reports_bad_token_ids():
    from multimodalsimilar_amd.bert import BertConfig, BertModel
    cfg = BertConfig(vocab_size=64, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = BertModel(cfg, seed=0).to(DEV).eval()
    ids = torch.randint(0, 64, (3, 32), device=DEV)
    with torch.no_grad():
        a = m(input_ids=ids).pooler_output
        b = m(input_ids=ids, position_ids=torch.arange(32, device=DEV)).pooler_output          # the default, spelled out
        c = m(input_ids=ids, position_ids=torch.arange(32, device=DEV).flip(0).unsqueeze(0)).pooler_output
    assert torch.equal(a, b) and (a - c).abs().max() > 1e-3
    m.check_indices()
    bad = ids.clone(); bad[0, 0] = 64
    with torch.no_grad():
        m(input_ids=bad)
    with pytest.raises(IndexError):
        m.check_indices()
    m.check_indices()          # the flag was cleared by the failed check


def test_colsum_and_l2norm():
    ops = _ops()
    x = rnd(1000, 264, seed=1).bfloat16()
    out = torch.zeros(264, device=DEV)
    ops.colsum(x, out)
    assert relerr(out, x.float().sum(0)) < 1e-4
    v = rnd(33, 512, seed=2, scale=3.0).requires_grad_(True)
    of = torch.zeros(33, 640, device=DEV); inv = torch.empty(33, device=DEV)
    ops.l2norm_fwd(v.detach(), of, None, 128, inv)
    ref = F.normalize(v, dim=1)
    assert torch.allclose(of[:, 128:], ref, atol=1e-6) and of[:, :128].abs().max() == 0
    g = rnd(33, 640, seed=3)
    ref.backward(g[:, 128:])
    dx = torch.empty(33, 512, device=DEV)
    ops.l2norm_bwd(v.detach(), inv, g, 128, dx)
    assert torch.allclose(dx, v.grad, rtol=1e-4, atol=1e-6)


def test_adamw_matches_golden(golden_dir):
    import numpy as np, os
    ops = _ops()
    from oracle import optim_ref
    d = np.load(os.path.join(golden_dir, "adamw_fc.npz"))
    lr0, warm, total = float(d["lr0"]), float(d["warmup"]), int(d["total"])
    shapes = [d[f"p0_{j}"].shape for j in range(3)]
    sizes = [int(np.prod(s)) for s in shapes]
    offs = [0, 40, 56]
    n = 72
    p = torch.zeros(n, device=DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    sh = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    for j in range(3):
        p[offs[j]:offs[j] + sizes[j]] = torch.from_numpy(d[f"p0_{j}"]).flatten().to(DEV)
    for t in range(total):
        g = torch.zeros(n, device=DEV)
        for j in range(3):
            g[offs[j]:offs[j] + sizes[j]] = torch.from_numpy(d[f"g{t}_{j}"]).flatten().to(DEV)
        ops.adamw_step(p, g, m, v, sh, optim_ref.linear_lr(lr0, t, warm, total), 0.9, 0.999, 1e-8, 0.01, t + 1)
        for j in range(3):
            ref = torch.from_numpy(d[f"p{t + 1}_{j}"]).flatten().to(DEV)
            assert torch.allclose(p[offs[j]:offs[j] + sizes[j]], ref, rtol=2e-5, atol=1e-6)
    assert torch.equal(sh, p.bfloat16())


@pytest.mark.parametrize("C,D,B", [(96, 1408, 8), (1000, 520, 24), (333, 72, 256)])
def test_head_weight_gradient_rowfix_epilogue(C, D, B):
    """The ArcFace weight gradient through the row-fix epilogue (mmsim_arcface_rowfix + mmsim_gemm_bf16 epilogue 5) against
    autograd through F.normalize(weight) (arcface.py:47) on the same bf16-rounded operands, accumulating into a non-zero buffer."""
    ops = _ops()
    w = rnd(C, D, scale=0.3, seed=1).requires_grad_(True)
    xh = F.normalize(rnd(B, D, seed=2)).bfloat16()
    dcos = rnd(B, C, scale=0.05, seed=3).bfloat16()
    wh_f = F.normalize(w)                                   # fp32; the kernels see its bf16 rounding
    wh = wh_f.detach().bfloat16()
    cos = (xh.float() @ wh.float().t()).contiguous()        # what the forward product returns
    ((xh.float() @ wh_f.t()) * dcos.float()).sum().backward()
    inv_w = (1.0 / w.detach().norm(dim=1)).contiguous()
    ldc = ops.round_up(C, 8)
    dcos_p = torch.zeros(B, ldc, dtype=torch.bfloat16, device=DEV); dcos_p[:, :C] = dcos
    cos_p = torch.zeros(B, ldc, device=DEV); cos_p[:, :C] = cos
    rowvec = torch.empty(2, C, device=DEV)
    ops.lib.arcface_rowfix(dcos_p.data_ptr(), cos_p.data_ptr(), ldc, inv_w.data_ptr(), rowvec.data_ptr(), B, C, ops._stream())
    assert relerr(rowvec[0], inv_w) == 0.0
    assert relerr(rowvec[1], (dcos.float() * cos).sum(0)) < 1e-5
    prev = rnd(C, D, scale=0.01, seed=4)
    out = prev.clone()
    ops.gemm(dcos_p[:, :C], xh, out, trans_a=True, b_kmajor=False, bias=rowvec, epilogue=ops.EPI_ROWFIX, aux_in=wh, accumulate=True)
    assert relerr(out - prev, w.grad) < 1e-2               # w_hat enters the correction bf16-rounded: 2^-9-class differences
    out2 = torch.full_like(prev, 7.0)                      # store form: previous contents ignored
    ops.gemm(dcos_p[:, :C], xh, out2, trans_a=True, b_kmajor=False, bias=rowvec, epilogue=ops.EPI_ROWFIX, aux_in=wh)
    assert torch.equal(out2, out - prev) or relerr(out2, out - prev) < 1e-5
    with pytest.raises(Exception):
        ops.gemm(dcos_p[:, :C], xh, out.bfloat16(), trans_a=True, b_kmajor=False, bias=rowvec, epilogue=ops.EPI_ROWFIX, aux_in=wh)


@pytest.mark.parametrize("R,D", [(37, 1408), (200, 2816), (5, 64), (9, 4096)])
def test_adamw_rows_l2norm_equals_adamw_then_normalize(R, D):
    """mmsim_adamw_rows_l2norm = mmsim_adamw_step followed by F.normalize of the updated rows (arcface.py:47)."""
    ops = _ops()
    p0, g = rnd(R, D, scale=0.1, seed=1), rnd(R, D, scale=0.01, seed=2)
    m0, v0 = rnd(R, D, scale=0.01, seed=3), rnd(R, D, scale=0.01, seed=4).abs() * 1e-3
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=3, grad_scale=0.5)
    pa, ma, va = p0.clone().view(-1), m0.clone().view(-1), v0.clone().view(-1)
    ops.adamw_step(pa, g.view(-1), ma, va, None, hp["lr"], hp["beta1"], hp["beta2"], hp["eps"], hp["weight_decay"], hp["step"], hp["grad_scale"])
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    wh = torch.empty(R, D, dtype=torch.bfloat16, device=DEV)
    inv = torch.empty(R, device=DEV)
    ops.adamw_rows_l2norm(pb, g, mb, vb, wh, inv, hp["lr"], hp["beta1"], hp["beta2"], hp["eps"], hp["weight_decay"], hp["step"], hp["grad_scale"])
    assert relerr(pb.view(-1), pa) < 1e-6 and relerr(mb.view(-1), ma) < 1e-6 and relerr(vb.view(-1), va) < 1e-6
    assert relerr(inv, 1.0 / pb.norm(dim=1)) < 1e-6
    assert relerr(wh, F.normalize(pb)) < 1e-2


def test_head_reuses_the_normalised_weights_only_while_they_are_current():
    """ArcMarginProduct keeps F.normalize(weight) from the optimiser launch / the previous forward; any torch-side write to the
    weight (version bump) or an update through the plain AdamW path must make the next forward renormalise."""
    from multimodalsimilar_amd import head as H
    from multimodalsimilar_amd.optim import FusedAdamW
    torch.manual_seed(0)
    mod = H.ArcMarginProduct(64, 40, m=0.3).to(DEV)
    x = rnd(6, 64, seed=5)
    label = torch.arange(6, device=DEV)
    opt = FusedAdamW(mod, lr=1e-2)
    def ref_cos():
        return F.normalize(x) @ F.normalize(mod.weight.detach()).t()
    for it in range(3):
        loss, _ = mod.forward_loss(x.clone().requires_grad_(True), label)
        loss.backward()
        opt.step(); opt.zero_grad()
        with torch.no_grad():
            assert relerr(mod.forward_test(x), ref_cos()) < 1e-2            # uses the w_hat the AdamW launch left
        assert mod._wh_key is not None
    with torch.no_grad():
        mod.weight.mul_(-1.0)                                               # torch-side write: version bump
        assert relerr(mod.forward_test(x), ref_cos()) < 1e-2
    old = H._FUSED_NORM
    try:
        H._FUSED_NORM = False                                               # plain AdamW path: must invalidate the cache
        loss, _ = mod.forward_loss(x.clone().requires_grad_(True), label)
        loss.backward()
        opt.step(); opt.zero_grad()
        assert mod._wh_key is None
        with torch.no_grad():
            assert relerr(mod.forward_test(x), ref_cos()) < 1e-2
    finally:
        H._FUSED_NORM = old



@pytest.mark.parametrize("M1,M2,N,K,sk", [(768, 256, 256, 1024, 2), (512, 512, 512, 640, 1), (3072, 1024, 1024, 2048, 4)])
def test_grouped_weight_gradient_pair(M1, M2, N, K, sk):
    """mmsim_gemm_bf16_wgrad_pair: two dY^T X products of one launch, accumulated into non-zero f32 buffers."""
    ops = _ops()
    a1, b1 = rnd(K, M1, seed=1).bfloat16(), rnd(K, N, seed=2).bfloat16()
    a2, b2 = rnd(K, M2, seed=3).bfloat16(), rnd(K, N, seed=4).bfloat16()
    c1, c2 = rnd(M1, N, seed=5), rnd(M2, N, seed=6)
    r1, r2 = c1 + a1.float().t() @ b1.float(), c2 + a2.float().t() @ b2.float()
    ops.gemm_wgrad_pair(a1, b1, c1, a2, b2, c2, sk)
    assert relerr(c1, r1) < 2e-3 and relerr(c2, r2) < 2e-3
    with pytest.raises(Exception):
        ops.gemm_wgrad_pair(a1[:, :100], b1, c1[:100], a2, b2, c2, sk)


def test_head_gradient_after_a_lazy_zero_grad_equals_the_eager_one():
    """optimizer.zero_grad(lazy=True) marks the head's gradient buffer instead of filling it; the dW product then overwrites.
    Gradients and updates must equal the eager zero_grad path bit for bit (C, D small: one tile path, no atomics), a second
    backward before the step accumulates, and a step with no backward in between sees zeros."""
    from multimodalsimilar_amd import head as H
    from multimodalsimilar_amd.optim import FusedAdamW
    x = rnd(8, 72, seed=5)
    label = torch.arange(8, device=DEV) * 3

    def run(lazy):
        torch.manual_seed(0)
        mod = H.ArcMarginProduct(72, 50, m=0.3).to(DEV)
        opt = FusedAdamW(mod, lr=1e-2)
        grads = []
        for it in range(3):
            loss, _ = mod.forward_loss(x.clone().requires_grad_(True), label)
            loss.backward()
            grads.append(mod.weight.grad.clone())
            opt.step(); opt.zero_grad(lazy=lazy)
            assert mod._flat.zero_pending == lazy
        return mod, opt, grads

    m0, o0, g0 = run(False)
    m1, o1, g1 = run(True)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    assert torch.equal(m0.weight.detach(), m1.weight.detach())
    # two backwards before a step: the first overwrites the stale buffer, the second accumulates
    loss, _ = m1.forward_loss(x.clone().requires_grad_(True), label); loss.backward()
    one = m1.weight.grad.clone()
    loss, _ = m1.forward_loss(x.clone().requires_grad_(True), label); loss.backward()
    assert relerr(m1.weight.grad, 2 * one) < 1e-6
    # a step right after a lazy zero_grad (no backward): the update sees zeros, not the stale gradient
    o1.zero_grad(lazy=True)
    w_before = m1.weight.detach().clone()
    o1.step()
    assert float(m1._flat.grad.abs().max()) == 0.0 and not m1._flat.zero_pending
    o0.zero_grad(); loss, _ = m0.forward_loss(x.clone().requires_grad_(True), label); loss.backward()
    loss, _ = m0.forward_loss(x.clone().requires_grad_(True), label); loss.backward()
    o0.zero_grad(); o0.step()
    assert torch.equal(m0.weight.detach(), m1.weight.detach()) and not torch.equal(w_before, m1.weight.detach())


@pytest.mark.parametrize("M,N,K,sk", [(4096, 1024, 4096, 4), (4096, 2048, 2048, 2)])
def test_weight_gradient_with_fused_bias_gradient(M, N, K, sk):
    """mmsim_gemm_bf16_wgrad_colsum: dW += dY^T X and dbias += column sums of dY from one pass over dY (the column sums ride on the
    weight-gradient MFMAs against an all-ones operand), against the two separate launches and fp32 torch; both accumulate."""
    from multimodalsimilar_amd import ops
    from multimodalsimilar_amd._lib import lib
    assert lib.gemm_bf16_wgrad_colsum_eligible(M, N, K, sk)
    dy = rnd(K, M, seed=1).bfloat16()
    x = rnd(K, N, seed=2).bfloat16()
    w0, b0 = rnd(M, N, seed=3), rnd(M, seed=4)
    dw, db = w0.clone(), b0.clone()
    ops.gemm_wgrad_colsum(dy, x, dw, db, sk)
    assert relerr(dw - w0, dy.float().t() @ x.float()) < 2e-3
    assert relerr(db - b0, dy.float().sum(0)) < 1e-4
    dw2, db2 = w0.clone(), b0.clone()
    ops.colsum(dy, db2)
    ops.gemm(dy, x, dw2, trans_a=True, b_kmajor=False, split_k=sk, accumulate=True)
    assert relerr(dw - w0, dw2 - w0) < 1e-4 and relerr(db - b0, db2 - b0) < 1e-4
    assert not lib.gemm_bf16_wgrad_colsum_eligible(M + 8, N, K, sk) and not lib.gemm_bf16_wgrad_colsum_eligible(256, 256, K, 1)
