"""Tensor-level wrappers over the C ABI (include/mmsim_hip.h).

Every wrapper validates dtype / device / contiguity / shape in Python and raises BEFORE anything is
launched, then passes raw device pointers and the current HIP stream to libmmsim_hip.so.
torch is used for memory and streams only.
"""
import os

import torch

from ._lib import lib, MmsimError

BF16 = torch.bfloat16
F16 = torch.float16
F32 = torch.float32
I64 = torch.int64

EPI_NONE, EPI_GELU, EPI_MUL_GELU_GRAD, EPI_ADD, EPI_TANH, EPI_ROWFIX, EPI_GELU_DGELU, EPI_MUL = 0, 1, 2, 3, 4, 5, 6, 7


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Side streams a model launches tower work on (multimodal_classifier.py: the image tower).  Whatever consumes the flat
# gradient buffers after backward -- the optimiser kernels, the gradient exchange -- first makes ITS stream wait for them:
# the towers write flat.grad with their own kernels (no AccumulateGrad node), so the ordering is stated here explicitly
# instead of being left to autograd's leaf-stream bookkeeping.
_side_streams = []


def register_side_stream(s):
    if all(s is not t for t in _side_streams):
        _side_streams.append(s)


def join_side_streams():
    """The current stream waits for everything enqueued so far on the registered side streams of its device."""
    if not _side_streams:
        return
    cur = torch.cuda.current_stream()
    for s in _side_streams:
        if s.device == cur.device and s != cur:
            cur.wait_stream(s)


# Device-side error flags (int32, set by kernels that validate indices: ArcFace labels, token / position / type ids).  They are
# read -- one host sync each -- where the caller asks for it: ArcMarginProduct.check_labels(), BertModel.check_indices(),
# check_device_flags() (all of them); the eager API path (module.forward) checks after every call like the reference raises.
_FLAGS = []


def register_error_flag(tensor, exc_type, message):
    import weakref
    _FLAGS.append((weakref.ref(tensor), exc_type, message))


def check_device_flags():
    live = []
    err = None
    for ref, exc_type, message in _FLAGS:
        t = ref()
        if t is None:
            continue
        live.append((ref, exc_type, message))
        if err is None and t.is_cuda and int(t.item()) != 0:
            t.zero_()
            err = exc_type(message)
    _FLAGS[:] = live
    if err is not None:
        raise err


def set_deterministic(on=True):
    """Deterministic (verification) mode of the HIP library: fixed-order reductions everywhere, no split-K.  Bit-identical
    results from run to run at a cost in step time (mmsim_set_deterministic).  MMSIM_DETERMINISTIC=1 turns it on at import."""
    lib.set_deterministic(int(bool(on)))


def is_deterministic():
    return bool(lib.get_deterministic())


def _chk(t, dtype, name, dims=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise MmsimError(f"{name}: tensor must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if dims is not None and t.dim() != dims:
        raise ValueError(f"{name}: expected {dims} dims, got shape {tuple(t.shape)}")
    if t.dim() >= 1 and t.numel() > 0 and t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")
    return t


def _p(t):
    return None if t is None else t.data_ptr()


def _ld(t):
    return t.stride(0) if t.dim() == 2 and t.shape[0] > 1 else t.shape[-1]


_SCRATCH = {}


def _scratch(device, nfloats):
    """Reusable fp32 scratch for per-block partial reductions (stream-ordered reuse: one stream per process)."""
    key = (device.type, device.index)
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(nfloats, 1 << 22), dtype=torch.float32, device=device)
        _SCRATCH[key] = t
    return t


def round_up(x, m):
    return (x + m - 1) // m * m


def alloc_2d(rows, cols, dtype, device, mult=8, zero=False):
    """[rows, cols] view of a buffer whose leading dimension is cols rounded up to `mult` (pad zeroed if asked)."""
    ld = round_up(cols, mult)
    buf = (torch.zeros if zero or ld != cols else torch.empty)((rows, ld), dtype=dtype, device=device)
    return buf[:, :cols] if ld != cols else buf


def gemm(a, b, c, *, trans_a=False, b_kmajor=True, bias=None, epilogue=EPI_NONE, aux_in=None, aux_out=None,
         alpha=1.0, split_k=1, accumulate=False):
    """c[M,N] = alpha * op(a) @ op(b) (+ bias) with fused epilogue; see mmsim_gemm_bf16.
    The element format follows the operand dtypes (mmsim_gemm_fmt): bf16 x bf16 (fmt 0); fp16 x fp16 -> fp16 / f32, forward layout
    (fmt 1: the image tower's forward products); bf16^T x fp16 -> f32 (fmt 2: its weight gradients, activation operand fp16)."""
    if a.dtype == F16 and b.dtype == F16:
        fmt, cdt = 1, F16
    elif a.dtype == BF16 and b.dtype == F16:
        fmt, cdt = 2, F32
    else:
        fmt, cdt = 0, BF16
    _chk(a, F16 if fmt == 1 else BF16, "gemm.a", 2); _chk(b, BF16 if fmt == 0 else F16, "gemm.b", 2)
    if c.dtype not in (cdt, F32):
        raise TypeError(f"gemm.c: {cdt} or f32 output for these operand dtypes")
    if fmt == 1 and (trans_a or not b_kmajor or split_k != 1):
        raise ValueError("gemm: fp16 operands are supported in the forward layout only (a [M,K], b [N,K], no split-K)")
    if fmt == 2 and (not trans_a or b_kmajor):
        raise ValueError("gemm: a bf16 gradient with an fp16 activation is the weight-gradient layout only (a [K,M], b [K,N])")
    _chk(c, c.dtype, "gemm.c", 2)
    M, N = c.shape
    K = a.shape[0] if trans_a else a.shape[1]
    am = a.shape[1] if trans_a else a.shape[0]
    bk, bn = (b.shape[1], b.shape[0]) if b_kmajor else (b.shape[0], b.shape[1])
    if am != M or bn != N or bk != K:
        raise ValueError(f"gemm: shape mismatch a{tuple(a.shape)} b{tuple(b.shape)} c{tuple(c.shape)} "
                         f"trans_a={trans_a} b_kmajor={b_kmajor}")
    if bias is not None:
        _chk(bias, F32, "gemm.bias")
        if bias.numel() != (2 * M if epilogue == EPI_ROWFIX else N):
            raise ValueError("gemm.bias: length must equal N (2*M row vectors for the row-fix epilogue)")
    ld_aux = 0
    for t, nm in ((aux_in, "gemm.aux_in"), (aux_out, "gemm.aux_out")):
        if t is not None:
            _chk(t, BF16, nm, 2)
            if tuple(t.shape) != (M, N):
                raise ValueError(f"{nm}: shape must equal the output shape")
            ld_aux = _ld(t)
    if fmt:
        lib.gemm_fmt(fmt, int(trans_a), int(b_kmajor), M, N, K, _p(a), _ld(a), _p(b), _ld(b), _p(c), _ld(c),
                     int(c.dtype == F32), _p(bias), epilogue, _p(aux_in), _p(aux_out), ld_aux, float(alpha), split_k,
                     int(accumulate), _stream())
        return c
    lib.gemm_bf16(int(trans_a), int(b_kmajor), M, N, K, _p(a), _ld(a), _p(b), _ld(b), _p(c), _ld(c),
                  int(c.dtype == F32), _p(bias), epilogue, _p(aux_in), _p(aux_out), ld_aux, float(alpha), split_k,
                  int(accumulate), _stream())
    return c


def wgrad_pair_eligible(M1, M2, N, K):
    return M1 % 256 == 0 and M2 % 256 == 0 and N % 256 == 0 and K % 64 == 0 and (M1 + M2) // 256 * (N // 256) >= 32


class gemm_group:
    """`with ops.gemm_group():` -- the (weight gradient, data gradient) products issued inside leave as one launch when both
    take the generic kernel (mmsim_gemm_group_begin / _end); MMSIM_GEMM_GROUP=0 turns the pairing off."""
    _on = os.environ.get("MMSIM_GEMM_GROUP", "1") != "0"

    def __enter__(self):
        if self._on:
            lib.gemm_group_begin()
        return self

    def __exit__(self, et, ev, tb):
        if self._on:
            if et is None:
                lib.gemm_group_end()
            else:            # do not mask the original error with a second one
                try:
                    lib.gemm_group_end()
                except Exception:
                    pass
        return False


def gemm_wgrad_pair(a1, b1, c1, a2, b2, c2, split_k):
    """c1[M1,N] += a1^T b1 and c2[M2,N] += a2^T b2 in one launch (a: [K, M], b: [K, N] bf16; c: f32); see mmsim_gemm_bf16_wgrad_pair."""
    for t, n in ((a1, "a1"), (b1, "b1"), (a2, "a2"), (b2, "b2")):
        _chk(t, BF16, "gemm_wgrad_pair." + n, 2)
    _chk(c1, F32, "gemm_wgrad_pair.c1", 2); _chk(c2, F32, "gemm_wgrad_pair.c2", 2)
    K, M1 = a1.shape
    M2, N = a2.shape[1], b1.shape[1]
    if a2.shape[0] != K or b1.shape[0] != K or b2.shape[0] != K or b2.shape[1] != N or tuple(c1.shape) != (M1, N) or tuple(c2.shape) != (M2, N):
        raise ValueError("gemm_wgrad_pair: shape mismatch")
    lib.gemm_bf16_wgrad_pair(M1, M2, N, K, _p(a1), _ld(a1), _p(b1), _ld(b1), _p(c1), _ld(c1), _p(a2), _ld(a2), _p(b2), _ld(b2), _p(c2),
                             _ld(c2), int(split_k), _stream())


def pick_split_k(M, N, K):
    """Split-K factor for wgrad-style products (few output tiles, long reduction).  Mirrors the tile choice of
    csrc/gemm_fast.hip: outputs with >= 32 tiles of 256x256 run the pipelined 256x256 kernel with ~192-256 blocks,
    smaller ones the 256x128 kernel with ~256-512 blocks (the split-K partial sums leave as fp32 atomics)."""
    mult = int(os.environ.get("MMSIM_SPLITK_MULT", "1"))
    if M % 256 == 0 and N % 256 == 0 and (M // 256) * (N // 256) >= 32:
        t, s = (M // 256) * (N // 256), 1
        while t * s < 192 and K // (s * 2) >= 512:
            s *= 2
        # any split count works (the K range is cut in multiples of 64): take the largest one that still fits the same
        # number of 256-CU rounds -- 48 tiles (the q|k|v weight gradient) run as 5 x 48 = 240 blocks instead of 4 x 48 = 192
        rounds = -(-(t * s) // 256)
        while (t * (s + 1)) <= rounds * 256 and K // (s + 1) >= 512 and os.environ.get("MMSIM_SPLITK_EXACT", "1") != "0":
            s += 1
        return s * mult
    # 128x128 generic kernel: every split adds a 64-KiB tile of fp32 ATOMICS (~1.3 TB/s chip-wide), so the count stops at ~192-384
    # blocks, not at two full rounds of 512 (tools/bench_imgemm2.py, late-stage weight gradients of the image tower: 39 tiles x 16
    # splits = 40 MB of atomics for a 1.8 MB gradient; halving the splits took the 20 late-stage blocks from 5.08 to 4.85 ms)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    s = 1
    # (outputs the pipelined 256x128 kernel takes -- M a multiple of 256, N of 128 -- keep its ~512 blocks: the text tower's
    # attention-output weight gradient on its own measured 720 TFLOP/s at 8 splits and 502 at 4)
    target = 512 if (M % 256 == 0 and N % 128 == 0) else int(os.environ.get("MMSIM_SPLITK_TARGET", "192"))
    while tiles * s < target and K // (s * 2) >= 512:
        s *= 2
    return s * mult


def attn_fwd(qkv, mask, ctx, lse, B, S, heads, H, dropout_p=0.0, seed=0, stream_id=0):
    _chk(qkv, BF16, "attn.qkv", 2); _chk(ctx, BF16, "attn.ctx", 2); _chk(lse, F32, "attn.lse")
    if mask is not None:
        _chk(mask, I64, "attn.mask", 2)
    if qkv.shape[0] != B * S or ctx.shape[0] != B * S or lse.numel() < B * heads * S:
        raise ValueError("attn_fwd: buffer shapes do not match B*S")
    lib.attn_fwd(_p(qkv), _ld(qkv), _p(mask), _p(ctx), _ld(ctx), _p(lse), B, S, heads, H, float(dropout_p), seed,
                 stream_id, _stream())


def attn_bwd(qkv, mask, ctx, dctx, lse, dqkv, B, S, heads, H, dropout_p=0.0, seed=0, stream_id=0, dbias=None):
    """dbias (fp32 [3H]): additionally dbias += dqkv.sum(0), out of the kernel (mmsim_attn_bwd_dbias)."""
    for t, n in ((qkv, "qkv"), (ctx, "ctx"), (dctx, "dctx"), (dqkv, "dqkv")):
        _chk(t, BF16, "attn_bwd." + n, 2)
    _chk(lse, F32, "attn_bwd.lse")
    if _ld(dqkv) != _ld(qkv) or _ld(dctx) != _ld(ctx):
        raise ValueError("attn_bwd: gradient buffers must share the leading dimensions of their primals")
    if dbias is not None:
        _chk(dbias, F32, "attn_bwd.dbias", 1)
        if dbias.shape[0] != 3 * H:
            raise ValueError("attn_bwd.dbias: length must be 3H")
        scr = _scratch(dqkv.device, B * 3 * H)
        lib.attn_bwd_dbias(_p(qkv), _ld(qkv), _p(mask), _p(ctx), _p(dctx), _ld(ctx), _p(lse), _p(dqkv), _p(dbias), B, S, heads, H,
                           float(dropout_p), seed, stream_id, _p(scr), scr.numel(), _stream())
        return
    lib.attn_bwd(_p(qkv), _ld(qkv), _p(mask), _p(ctx), _p(dctx), _ld(ctx), _p(lse), _p(dqkv), B, S, heads, H,
                 float(dropout_p), seed, stream_id, _stream())


def _embed_tables(ids, tts, pids, word, pos, typ, err, B, S, name):
    _chk(ids, I64, name + ".ids")
    for t, n in ((tts, "token_types"), (pids, "position_ids")):
        if t is not None:
            _chk(t, I64, f"{name}.{n}")
            if t.numel() != B * S or not t.is_contiguous():
                raise ValueError(f"{name}.{n}: must hold B*S contiguous indices")
    if err.dtype != torch.int32 or not err.is_cuda:
        raise TypeError(f"{name}.err: int32 device flag required")
    if ids.numel() != B * S or not ids.is_contiguous():
        raise ValueError(f"{name}: ids must hold B*S contiguous tokens")
    if pids is None and pos.shape[0] < S:
        raise ValueError(f"{name}: the position table must cover S")
    if typ.shape[0] > 2:
        raise ValueError(f"{name}: at most two token types are supported")
    return word.shape[0], typ.shape[0], pos.shape[0]


def embed_ln_fwd(ids, tts, word, pos, typ, gamma, beta, out, B, S, H, eps, err, dropout_p=0.0, seed=0, stream_id=0, pids=None):
    _chk(out, BF16, "embed.out", 2)
    for t, n in ((word, "word"), (pos, "pos"), (typ, "type"), (gamma, "gamma"), (beta, "beta")):
        _chk(t, F32, "embed." + n)
    V, TV, P = _embed_tables(ids, tts, pids, word, pos, typ, err, B, S, "embed")
    lib.embed_ln_fwd(_p(ids), _p(tts), _p(pids), _p(word), _p(pos), _p(typ), _p(gamma), _p(beta), _p(out), B, S, H, V, TV, P,
                     _p(err), eps, float(dropout_p), seed, stream_id, _stream())


def embed_ln_bwd(dout, ids, tts, word, pos, typ, gamma, dword, dpos, dtype_, dgamma, dbeta, B, S, H, eps, err,
                 dropout_p=0.0, seed=0, stream_id=0, pids=None):
    _chk(dout, BF16, "embed_bwd.dout", 2)
    for t, n in ((dword, "dword"), (dpos, "dpos"), (dtype_, "dtype"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _chk(t, F32, "embed_bwd." + n)
    V, TV, P = _embed_tables(ids, tts, pids, word, pos, typ, err, B, S, "embed_bwd")
    scr = _scratch(dout.device, lib.embed_ln_bwd_scratch_floats(B, S, H))      # one slab row [4][H] per workgroup (the library's count)
    lib.embed_ln_bwd2(_p(dout), _p(ids), _p(tts), _p(pids), _p(word), _p(pos), _p(typ), _p(gamma), _p(dword), _p(dpos),
                     _p(dtype_), _p(dgamma), _p(dbeta), B, S, H, V, TV, P, _p(err), eps, float(dropout_p), seed, stream_id,
                     _p(scr), scr.numel(), _stream())


def add_ln_fwd(t, resid, gamma, beta, y, h, mean, rstd, eps, dropout_p=0.0, seed=0, stream_id=0):
    for x, n in ((t, "t"), (resid, "resid"), (y, "y"), (h, "h")):
        _chk(x, BF16, "add_ln." + n, 2)
        if not x.is_contiguous() or x.shape != t.shape:
            raise ValueError(f"add_ln.{n}: must be contiguous [M,H]")
    M, H = t.shape
    lib.add_ln_fwd(_p(t), _p(resid), _p(gamma), _p(beta), _p(y), _p(h), _p(mean), _p(rstd), M, H, eps,
                   float(dropout_p), seed, stream_id, _stream())


def ln_bwd(dh_a, dh_b, y, mean, rstd, gamma, dy, dt, dgamma, dbeta, dbias, dropout_p=0.0, seed=0, stream_id=0):
    for x, n in ((dh_a, "dh_a"), (y, "y"), (dy, "dy")):
        _chk(x, BF16, "ln_bwd." + n, 2)
        if not x.is_contiguous():
            raise ValueError(f"ln_bwd.{n}: must be contiguous")
    M, H = y.shape
    scr = _scratch(y.device, 2048 * 3 * H)
    lib.ln_bwd(_p(dh_a), _p(dh_b), _p(y), _p(mean), _p(rstd), _p(gamma), _p(dy), _p(dt), _p(dgamma), _p(dbeta),
               _p(dbias), M, H, float(dropout_p), seed, stream_id, _p(scr), scr.numel(), _stream())


def gemm_wgrad_colsum(dy, x, dw, dbias, split_k):
    """dw[M,N] += dy^T x  and  dbias[M] += column sums of dy, one pass over dy (dy: [K, M], x: [K, N] bf16; dw, dbias f32); falls back
    to the two separate launches when the shape does not take the pipelined kernel (mmsim_gemm_bf16_wgrad_colsum_eligible)."""
    _chk(dy, BF16, "gemm_wgrad_colsum.dy", 2); _chk(x, BF16, "gemm_wgrad_colsum.x", 2)
    _chk(dw, F32, "gemm_wgrad_colsum.dw", 2); _chk(dbias, F32, "gemm_wgrad_colsum.dbias", 1)
    K, M = dy.shape
    N = x.shape[1]
    if x.shape[0] != K or tuple(dw.shape) != (M, N) or dbias.numel() != M:
        raise ValueError("gemm_wgrad_colsum: shape mismatch")
    if _WGRAD_COLSUM and lib.gemm_bf16_wgrad_colsum_eligible(M, N, K, int(split_k)):
        lib.gemm_bf16_wgrad_colsum(M, N, K, _p(dy), _ld(dy), _p(x), _ld(x), _p(dw), _ld(dw), _p(dbias), int(split_k), _stream())
    else:
        colsum(dy, dbias)
        gemm(dy, x, dw, trans_a=True, b_kmajor=False, split_k=split_k, accumulate=True)


# MMSIM_WGRAD_COLSUM=0: bias gradient of intermediate.dense by the separate column-sum pass (A/B switch)
_WGRAD_COLSUM = os.environ.get("MMSIM_WGRAD_COLSUM", "1") != "0"


def colsum(x, out):
    _chk(x, BF16, "colsum.x", 2); _chk(out, F32, "colsum.out", 1)
    lib.colsum_bf16(_p(x), _ld(x), _p(out), x.shape[0], x.shape[1], _stream())


def l2norm_fwd(x, out_f32, out_bf16, col_off, inv_norm, eps=1e-12, post_scale=1.0):
    if x.dtype not in (BF16, F32):
        raise TypeError("l2norm_fwd.x: bf16 or f32")
    _chk(x, x.dtype, "l2norm_fwd.x", 2)
    o = out_f32 if out_f32 is not None else out_bf16
    R, D = x.shape
    lib.l2norm_fwd(_p(x), int(x.dtype == BF16), _ld(x), _p(out_f32), _p(out_bf16), _ld(o), col_off, _p(inv_norm), R, D,
                   eps, float(post_scale), _stream())


def l2norm_bwd(x, inv_norm, dxh, col_off, dx, pre_scale=1.0, accumulate=False):
    _chk(x, x.dtype, "l2norm_bwd.x", 2); _chk(dxh, F32, "l2norm_bwd.dxh", 2); _chk(dx, F32, "l2norm_bwd.dx", 2)
    R, D = x.shape
    lib.l2norm_bwd(_p(x), int(x.dtype == BF16), _ld(x), _p(inv_norm), _p(dxh), _ld(dxh), col_off, _p(dx), _ld(dx), R, D,
                   float(pre_scale), int(accumulate), _stream())


def cast_to_bf16(x, y):
    _chk(x, F32, "cast.x"); _chk(y, BF16, "cast.y")
    if x.numel() != y.numel() or not x.is_contiguous() or not y.is_contiguous():
        raise ValueError("cast: contiguous tensors of equal size")
    lib.cast_f32_to_bf16(_p(x), _p(y), x.numel(), _stream())


def cast_to_f16(x, y):
    _chk(x, F32, "cast.x"); _chk(y, F16, "cast.y")
    if x.numel() != y.numel() or not x.is_contiguous() or not y.is_contiguous():
        raise ValueError("cast: contiguous tensors of equal size")
    lib.cast_f32_to_f16(_p(x), _p(y), x.numel(), _stream())


def cast_to_f32(x, y):
    _chk(x, BF16, "cast.x"); _chk(y, F32, "cast.y")
    if x.numel() != y.numel() or not x.is_contiguous() or not y.is_contiguous():
        raise ValueError("cast: contiguous tensors of equal size")
    lib.cast_bf16_to_f32(_p(x), _p(y), x.numel(), _stream())


def adamw_step(p, g, m, v, shadow, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, dev_hyper=None, shadow16=None):
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, F32, "adamw." + n, 1)
        if t.numel() != p.numel():
            raise ValueError("adamw: buffers must have equal length")
    if shadow is not None:
        _chk(shadow, BF16, "adamw.shadow", 1)
    if shadow16 is not None:
        _chk(shadow16, F16, "adamw.shadow16", 1)
    lib.adamw_step2(_p(p), _p(g), _p(m), _p(v), _p(shadow), _p(shadow16), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                    float(weight_decay), int(step), float(grad_scale), _p(dev_hyper), _stream())


def adamw_rows_l2norm(p2d, g2d, m2d, v2d, w_hat, inv_norm, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, l2_eps=1e-12,
                      dev_hyper=None):
    """AdamW on a [R, D] weight matrix that also leaves F.normalize(weight) (bf16) and 1/||row|| for the next forward."""
    R, D = p2d.shape
    for t, n in ((p2d, "p"), (g2d, "g"), (m2d, "m"), (v2d, "v")):
        _chk(t, F32, "adamw_rows." + n, 2)
        if tuple(t.shape) != (R, D) or not t.is_contiguous():
            raise ValueError("adamw_rows: p, g, m, v must be contiguous [R, D]")
    _chk(w_hat, BF16, "adamw_rows.w_hat", 2); _chk(inv_norm, F32, "adamw_rows.inv_norm", 1)
    if tuple(w_hat.shape) != (R, D) or not w_hat.is_contiguous() or inv_norm.numel() != R:
        raise ValueError("adamw_rows: w_hat [R, D] contiguous and inv_norm [R] required")
    lib.adamw_rows_l2norm(_p(p2d), _p(g2d), _p(m2d), _p(v2d), _p(w_hat), _p(inv_norm), R, D, float(lr), float(beta1), float(beta2),
                          float(eps), float(weight_decay), int(step), float(grad_scale), float(l2_eps), _p(dev_hyper), _stream())


if os.environ.get("MMSIM_DETERMINISTIC", "0") == "1":
    set_deterministic(True)
