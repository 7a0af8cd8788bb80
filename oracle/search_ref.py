"""CPU restatement of the reference's similarity search (TEST INFRASTRUCTURE ONLY, never imported by the product path).

Reference: nlp_infer.py:139-152 (also daodian_infer.py:225-230, 295-302):
    normalize_L2(x); index = faiss.IndexFlat(d, faiss.METRIC_INNER_PRODUCT); index.add(x); D, I = index.search(x, k)
faiss (no pinned version in the reference, NOT installed in this image, sources not under /root/reference) is the third-party
dependency that holds the algorithm; its published behaviour for IndexFlat / METRIC_INNER_PRODUCT is restated here: exact fp32
inner products against every stored vector, the k largest per query in descending order, -1 / -inf padding when fewer than k
vectors are stored.  PARITY UNPINNED: the reference's tests hold no vectors for this step and faiss cannot be run here; the
order among exactly equal scores (implementation-defined in faiss) is fixed to ascending index.
"""
import numpy as np


def normalize_l2(x):
    """faiss.normalize_L2: each row divided by its Euclidean norm (zero rows are left untouched)."""
    x = np.array(x, dtype=np.float32, copy=True)
    n = np.sqrt((x.astype(np.float64) ** 2).sum(1))
    nz = n > 0
    x[nz] = (x[nz] / n[nz, None]).astype(np.float32)
    return x


def search_inner_product(queries, database, k, normalize=True):
    q = normalize_l2(queries) if normalize else np.asarray(queries, np.float32)
    d = normalize_l2(database) if normalize else np.asarray(database, np.float32)
    s = q.astype(np.float64) @ d.astype(np.float64).T
    nq, n = s.shape
    D = np.full((nq, k), -np.inf, np.float32)
    I = np.full((nq, k), -1, np.int64)
    kk = min(k, n)
    order = np.argsort(-s, axis=1, kind="stable")[:, :kk]          # stable: equal scores by ascending index
    D[:, :kk] = np.take_along_axis(s, order, 1).astype(np.float32)
    I[:, :kk] = order
    return D, I


def search_l2(queries, database, k):
    """faiss.IndexFlatL2 (multimodal_infer.py:140-145): exact squared Euclidean distances to every stored vector, the k smallest
    per query in ascending order, -1 / +inf padding; equal distances by ascending index (implementation-defined in faiss)."""
    q = np.asarray(queries, np.float64)
    d = np.asarray(database, np.float64)
    s = (q ** 2).sum(1)[:, None] + (d ** 2).sum(1)[None, :] - 2.0 * q @ d.T
    nq, n = s.shape
    D = np.full((nq, k), np.inf, np.float32)
    I = np.full((nq, k), -1, np.int64)
    kk = min(k, n)
    order = np.argsort(s, axis=1, kind="stable")[:, :kk]
    D[:, :kk] = np.maximum(np.take_along_axis(s, order, 1), 0.0).astype(np.float32)
    I[:, :kk] = order
    return D, I
