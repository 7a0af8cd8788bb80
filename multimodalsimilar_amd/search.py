"""Exhaustive (flat) top-k search on the GPU, inner product and L2: the step that follows the model in the reference's
inference jobs -- ``faiss.normalize_L2(x); index = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT); index.add(x);
D, I = index.search(x, k)`` (nlp_infer.py:139-152, daodian_infer.py:225-230, 295-302).

``topk_inner_product(queries, database, k)`` returns ``(D, I)`` with faiss's meaning: per query the k largest inner
products in descending order and the database row indices they belong to (int64; -1 / -inf where the database has fewer
than k rows).  Equal scores are ordered by ascending index.  Scores are fp32 inner products computed on the MFMA units from
bf16 hi/lo splits of the operands (error ~2^-16 relative), in database chunks so the [nq, N] score matrix never exists.
``topk_l2(queries, database, k)`` is ``faiss.IndexFlatL2(d); index.add(x); D, I = index.search(x, k)`` of the two-tower
inference job (multimodal_infer.py:140-145): the k smallest SQUARED Euclidean distances in ascending order.  It runs on the same
kernels through ||q - d||^2 = ||q||^2 - (2 q.d - ||d||^2): one augmented inner product per pair, ranked descending.
There is no CPU fallback.
"""
import torch

from . import ops
from ._lib import lib

_CHUNK_DB = 16384        # database rows per GEMM
_CHUNK_Q = 16384         # query rows per pass (bounds the fp32 score buffer: 16384 x 16384 x 4 B = 1 GiB)


def _split(x, db_side, normalize):
    R, D = x.shape
    x = x.contiguous().float()
    if normalize == "l2-aug":
        # augmented operands of the L2 search: query [q, 1, 0..], database [2 d, -||d||^2, 0..]  (width D + 8: 16-byte rows)
        aug = torch.zeros(R, D + 8, dtype=torch.float32, device=x.device)
        if db_side:
            aug[:, :D] = 2.0 * x
            aug[:, D] = -(x.double() ** 2).sum(1).float()
        else:
            aug[:, :D] = x
            aug[:, D] = 1.0
        x, D, normalize = aug, D + 8, False
    if normalize:
        xn = torch.empty_like(x)
        ops.l2norm_fwd(x, xn, None, 0, None)           # faiss.normalize_L2: rows / ||row||_2
        x = xn
    out = ops.alloc_2d(R, 3 * D, torch.bfloat16, x.device)
    lib.split_bf16_cat(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), R, D, int(db_side), 1.0, ops._stream())
    return out


def topk_inner_product(queries, database, k, normalize=True):
    """queries [nq, D], database [N, D] (fp32 / bf16 GPU tensors) -> (scores fp32 [nq, k], indices int64 [nq, k])."""
    if not (queries.is_cuda and database.is_cuda):
        raise ops.MmsimError("topk_inner_product: tensors must live on the GPU; there is no CPU path")
    if queries.dim() != 2 or database.dim() != 2 or queries.shape[1] != database.shape[1]:
        raise ValueError("topk_inner_product: expected [nq, D] and [N, D]")
    if not 1 <= k <= 64:
        raise ValueError("topk_inner_product: 1 <= k <= 64")
    nq, D = queries.shape
    N = database.shape[0]
    db = _split(database, 1, normalize)
    dev = queries.device
    best_v = torch.empty(nq, k, dtype=torch.float32, device=dev)
    best_i = torch.empty(nq, k, dtype=torch.int64, device=dev)
    s = ops._stream()
    for q0 in range(0, nq, _CHUNK_Q):
        q1 = min(nq, q0 + _CHUNK_Q)
        qs = _split(queries[q0:q1], 0, normalize)
        for c0 in range(0, N, _CHUNK_DB):
            c1 = min(N, c0 + _CHUNK_DB)
            sc = ops.alloc_2d(q1 - q0, c1 - c0, torch.float32, dev)
            ops.gemm(qs, db[c0:c1], sc)                                  # fp32 scores of this (query, database) block
            lib.topk_merge(sc.data_ptr(), sc.stride(0), q1 - q0, c1 - c0, c0, k, best_v[q0:q1].data_ptr(),
                           best_i[q0:q1].data_ptr(), int(c0 == 0), s)
    return best_v, best_i


def topk_l2(queries, database, k):
    """queries [nq, D], database [N, D] -> (squared L2 distances fp32 [nq, k] ascending, indices int64 [nq, k]); faiss
    IndexFlatL2 semantics (multimodal_infer.py:140-145), +inf / -1 where the database has fewer than k rows."""
    score, idx = topk_inner_product(queries, database, k, normalize="l2-aug")
    qn = (queries.float().double() ** 2).sum(1, keepdim=True).float()
    dist = qn - score                       # -inf scores (padding) become +inf distances
    return torch.where(idx >= 0, dist.clamp_min(0.0), torch.full_like(dist, float("inf"))), idx
