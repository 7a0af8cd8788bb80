"""Similarity search on the MI355X against the CPU restatement of faiss IndexFlat / METRIC_INNER_PRODUCT
(oracle/search_ref.py; parity unpinned w.r.t. faiss, which is not installed)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _check(D, I, Dr, Ir, tol=4e-5):
    D, I = D.cpu().numpy(), I.cpu().numpy()
    fin = np.isfinite(Dr)
    assert np.array_equal(np.isfinite(D), fin) and np.array_equal(I[~fin], Ir[~fin])          # -inf / -1 padding
    assert np.abs(D[fin] - Dr[fin]).max() < tol
    mism = (I != Ir) & fin
    if mism.any():          # only genuine near-ties may swap: the oracle's scores of the two candidates differ by < 2 tol
        rows, cols = np.nonzero(mism)
        for r, c in zip(rows, cols):
            assert abs(Dr[r, c] - D[r, c]) < 2 * tol, (r, c, I[r, c], Ir[r, c])
    assert mism.mean() < 0.01


@pytest.mark.parametrize("nq,N,D,k", [(300, 1000, 64, 13), (256, 4096, 768, 13), (70, 50, 32, 64), (5, 3, 16, 8)])
def test_topk_inner_product_matches_oracle(nq, N, D, k):
    from multimodalsimilar_amd import search
    from oracle import search_ref
    g = torch.Generator().manual_seed(nq + N)
    q = torch.randn(nq, D, generator=g)
    x = torch.randn(N, D, generator=g)
    Dr, Ir = search_ref.search_inner_product(q.numpy(), x.numpy(), k)
    Dm, Im = search.topk_inner_product(q.to(DEV), x.to(DEV), k)
    _check(Dm, Im, Dr, Ir)


def test_self_search_ties_and_chunking(monkeypatch):
    """The reference's use: every vector against the whole set (nlp_infer.py:152).  Exact duplicates tie (ascending index);
    small chunks force the running lists through many merges."""
    from multimodalsimilar_amd import search
    from oracle import search_ref
    monkeypatch.setattr(search, "_CHUNK_DB", 96)
    monkeypatch.setattr(search, "_CHUNK_Q", 128)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(500, 48, generator=g)
    x[100] = x[7]; x[101] = x[7]; x[499] = x[0]                    # duplicates -> equal scores
    Dr, Ir = search_ref.search_inner_product(x.numpy(), x.numpy(), 13)
    Dm, Im = search.topk_inner_product(x.to(DEV), x.to(DEV), 13)
    _check(Dm, Im, Dr, Ir)
    Im = Im.cpu().numpy()
    assert list(Im[7, :3]) == [7, 100, 101] and list(Im[100, :3]) == [7, 100, 101]      # ties by ascending index
    assert abs(Dm[7, 0].item() - 1.0) < 1e-4


def test_search_rejects_cpu_and_large_k():
    from multimodalsimilar_amd import search, MmsimError
    with pytest.raises(MmsimError):
        search.topk_inner_product(torch.randn(4, 8), torch.randn(4, 8), 2)
    with pytest.raises(ValueError):
        search.topk_inner_product(torch.randn(4, 8, device=DEV), torch.randn(4, 8, device=DEV), 65)


@pytest.mark.parametrize("nq,N,D,k", [(200, 1500, 64, 13), (64, 3000, 2560, 13), (9, 5, 24, 8)])
def test_topk_l2_matches_oracle(nq, N, D, k):
    """faiss.IndexFlatL2 as the two-tower inference job uses it (multimodal_infer.py:140-145: d = 2560, k = 13, every vector
    against the whole set): squared distances ascending, self-match first at distance ~0."""
    from multimodalsimilar_amd import search
    from oracle import search_ref
    g = torch.Generator().manual_seed(nq * 7 + N)
    x = torch.randn(N, D, generator=g)
    x = torch.nn.functional.normalize(x[:, :D // 2], dim=1).repeat(1, 2) if D == 2560 else x      # two unit halves, as the model emits
    q = x[:nq].clone() if N >= nq else torch.randn(nq, D, generator=g)
    Dr, Ir = search_ref.search_l2(q.numpy(), x.numpy(), k)
    Dm, Im = search.topk_l2(q.to(DEV), x.to(DEV), k)
    Dm, Im = Dm.cpu().numpy(), Im.cpu().numpy()
    fin = np.isfinite(Dr)
    assert np.array_equal(np.isfinite(Dm), fin) and np.array_equal(Im[~fin], Ir[~fin])
    scale = max(1.0, float(np.abs(Dr[fin]).max()))
    assert np.abs(Dm[fin] - Dr[fin]).max() < 2e-4 * scale
    mism = (Im != Ir) & fin
    for r, c in zip(*np.nonzero(mism)):      # only near-ties may swap
        assert abs(Dr[r, c] - Dm[r, c]) < 4e-4 * scale
    assert mism.mean() < 0.02
    if N >= nq:
        assert (Im[:, 0] == np.arange(nq)).mean() > 0.99 and Dm[:, 0].max() < 2e-4 * scale
