"""Class-sharded ArcFace head for data parallelism with very many classes (BASELINE config 5: 8 ranks,
1 000 000 classes, D = 2816).

A replicated head would all-reduce an 11.3 GB fp32 weight gradient per step (~129 ms on a ring of 153 GB/s xGMI links) and
run AdamW over 1 M x 2816 parameters on every rank.  Here rank r owns the class rows [r C/N, (r+1) C/N) of the weight and
its optimiser state; per step the ranks exchange only

    forward   all-gather of the embeddings [B_loc, D] and labels            (23 MB at N = 8, B_loc = 256, D = 2816)
              all-gather of per-row partial softmax statistics [B_glob, 4]   (a few KB)
    backward  reduce-scatter of dX [B_glob, D] -> [B_loc, D]                 (23 MB in, 2.9 MB out)

and every rank computes cosines / margins / softmax terms for ALL rows of the global batch against ITS classes: the same
FLOPs per rank as the replicated head (B_loc x C), 1/N of its weight traffic and optimiser work, and no gradient
all-reduce for the head at all (a shard's gradient is complete locally).

Semantics kept (reference: one softmax over all classes, mean over the batch; under data parallelism the update is the
gradient of the GLOBAL-batch mean, nlp_classifier_train_daodian_v2_dist.py:139-144): the module returns the mean loss of the
LOCAL rows, its backward follows the towers' convention (each rank differentiates its local mean; the exchange sums over
ranks; FusedAdamW folds in 1/world) -- so dX is d(local mean)/dx and a shard's dW is the sum over all ranks' local means.
Equivalence with the replicated head is tested with world_size 2 on gloo (tests/test_sharded_head_gloo.py, CPU, local
maths from the oracle) and on the GPU box (tests/test_gpu_sharded_head.py, the HIP kernels).
"""
import math

import torch
import torch.distributed as dist

from . import ops
from .head import ArcMarginProduct


def shard_range(num_classes, world, rank):
    per = -(-num_classes // world)
    c0 = min(rank * per, num_classes)
    return c0, min(per, num_classes - c0)


class ShardedArcMarginProduct(ArcMarginProduct):
    """ArcMarginProduct whose ``weight`` holds only this rank's class rows.  ``out_feature`` stays the TOTAL class count;
    ``class_offset`` / ``local_classes`` describe the shard.  API: ``forward_loss`` (training), ``forward_test`` (all-gathered
    cosines), ``update_m``; the literal ``forward`` (full margin logits) is not offered -- nothing holds all columns."""

    dp_exchange = False       # GradientExchange must not all-reduce this gradient: it is complete on its owner

    def __init__(self, in_feature=128, out_feature=10575, s=64.0, m=0.40, easy_margin=False, process_group=None, seed=0,
                 full_weight=None):
        on = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size(process_group) if on else 1
        rank = dist.get_rank(process_group) if on else 0
        c0, cl = shard_range(out_feature, world, rank)
        if cl <= 0:
            raise ValueError(f"ShardedArcMarginProduct: rank {rank} of {world} would own no class of {out_feature}")
        if out_feature >= 1 << 24:
            raise ValueError("ShardedArcMarginProduct: class indices travel as fp32 columns of the gathered rows (exact below 2^24)")
        super().__init__(in_feature, cl, s=s, m=m, easy_margin=easy_margin)
        self.pg, self.world, self.rank = process_group, world, rank
        self.class_offset, self.local_classes, self.total_classes = c0, cl, out_feature
        self.out_feature = out_feature
        with torch.no_grad():
            w = self._flat.view("weight")
            if full_weight is not None:
                w.copy_(full_weight[c0:c0 + cl])
            else:       # xavier_uniform_ of the FULL [C, D] matrix (arcface.py:25): the bound depends on the total fan
                bound = math.sqrt(6.0 / (out_feature + in_feature))
                g = torch.Generator().manual_seed(seed * 1000003 + self.rank)
                w.copy_((torch.rand(w.shape, generator=g) * 2 - 1) * bound)

    # the base class sizes its buffers from out_feature: inside its kernels' calls the head is "a head with local_classes classes"
    class _Local:
        def __init__(self, mod):
            self.mod = mod

        def __enter__(self):
            self.mod.out_feature = self.mod.local_classes

        def __exit__(self, *a):
            self.mod.out_feature = self.mod.total_classes

    def adamw_row_buffers(self):
        with self._Local(self):
            return super().adamw_row_buffers()

    # ---- collectives (gloo rehearsal with tensors on a GPU stages through the host, as dist.GradientExchange does)
    def _gather(self, t):
        """[...] -> [world, ...] of every rank's tensor."""
        if self.world == 1:
            return t.unsqueeze(0)
        stage = t.is_cuda and dist.get_backend(self.pg) == "gloo"
        src = t.cpu() if stage else t.contiguous()
        if dist.get_backend(self.pg) == "nccl":
            # RCCL: ONE collective into one [world, ...] tensor (no list of per-rank outputs, no torch.stack copy afterwards)
            out = torch.empty((self.world,) + tuple(src.shape), dtype=src.dtype, device=src.device)
            dist.all_gather_into_tensor(out, src, group=self.pg)
            return out
        outs = [torch.empty_like(src) for _ in range(self.world)]
        dist.all_gather(outs, src, group=self.pg)
        out = torch.stack(outs)
        return out.to(t.device) if stage else out

    def _reduce_scatter_rows(self, full, rows):
        """sum over ranks of full [world * rows, D] -> this rank's [rows, D]."""
        if self.world == 1:
            return full
        if dist.get_backend(self.pg) == "gloo":          # no reduce_scatter on gloo: all-reduce and slice
            h = full.cpu() if full.is_cuda else full.clone()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
            return h[self.rank * rows:(self.rank + 1) * rows].to(full.device).contiguous()
        out = torch.empty(rows, full.shape[1], dtype=full.dtype, device=full.device)
        dist.reduce_scatter_tensor(out, full.contiguous(), op=dist.ReduceOp.SUM, group=self.pg)
        return out

    # ---- local maths on the HIP kernels (tests substitute the oracle's to run the exchange logic on CPU)
    def _local_cosines(self, X):
        with self._Local(self):
            return self._cosines(X)

    def _partial_stats(self, cos, st, Y):
        Bg = st["B"]
        stats = torch.empty(Bg, 4, dtype=torch.float32, device=cos.device)
        arg = torch.empty(Bg, dtype=torch.int64, device=cos.device)
        ops.lib.arcface_ce_partial(cos.data_ptr(), st["ldc"], Y.data_ptr(), stats.data_ptr(), arg.data_ptr(), Bg, self.local_classes,
                                   self.class_offset, self.total_classes, self.s, self.m, int(self.easy_margin),
                                   self._err_flag().data_ptr(), ops._stream())
        return stats, arg

    def _local_dcos(self, cos, st, Y, lse, row_scale):
        dcos = self._buf("dcos", (st["B"], st["ldc"]), torch.bfloat16)
        ops.lib.arcface_dcos_from_lse(cos.data_ptr(), st["ldc"], Y.data_ptr(), lse.data_ptr(), row_scale.data_ptr(), dcos.data_ptr(),
                                      st["B"], self.local_classes, self.class_offset, self.s, self.m, int(self.easy_margin),
                                      ops._stream())
        return dcos

    def _local_backward(self, st, dcos, cos):
        with self._Local(self):
            return self._backward_from_dcos(st, dcos, cos)

    # ---- API
    def forward(self, x, label):
        raise NotImplementedError("ShardedArcMarginProduct: the full [B, C] margin logits live on no rank; use forward_loss "
                                  "(training) or forward_test (all-gathered cosines)")

    def forward_loss(self, x, label, want_argmax=True):
        """-> (mean cross-entropy of this rank's rows, argmax over ALL classes for this rank's rows)."""
        return _ShardedLossFn.apply(x, self.weight, self, label)

    def predict(self, x):
        """(max cosine, argmax class) over ALL classes for this rank's rows -- the evaluation the train loop needs
        (multimodal_classifier_train.py:215-224 takes argmax of forward_test), exchanged as two numbers per row and rank."""
        with torch.no_grad():
            B = x.shape[0]
            X = self._gather(x.contiguous().float()).reshape(self.world * B, -1)
            cos, _ = self._local_cosines(X)
            v, i = cos[:, :self.local_classes].max(1)
            V, I = self._gather(v), self._gather(i + self.class_offset)
            r = V.argmax(0)
            rows = slice(self.rank * B, (self.rank + 1) * B)
            return V.gather(0, r.unsqueeze(0))[0][rows], I.gather(0, r.unsqueeze(0))[0][rows]

    def forward_test(self, x):
        """Cosines against all classes [B_loc, C] (arcface.py:65-67): every rank's columns, all-gathered.  B_glob x C floats
        travel to every rank -- an evaluation convenience for moderate C; use ``predict`` when only the argmax is needed."""
        with torch.no_grad():
            B = x.shape[0]
            X = self._gather(x.contiguous().float()).reshape(self.world * B, -1)
            cos, _ = self._local_cosines(X)
            mine = cos[self.rank * B:(self.rank + 1) * B]
            if self.world == 1:
                return mine[:, :self.local_classes].clone()
            per = shard_range(self.total_classes, self.world, 0)[1]
            # every rank needs ITS rows against every shard: exchange [world, B, per] blocks
            blocks = cos.new_zeros(self.world, B, per)
            blocks[:, :, :self.local_classes] = cos[:, :self.local_classes].reshape(self.world, B, self.local_classes)
            got = self._gather(blocks)                        # [src rank, dst rank, B, per]
            return torch.cat([got[r, self.rank, :, :shard_range(self.total_classes, self.world, r)[1]] for r in range(self.world)], 1)


def combine_row_stats(S, A):
    """S [world, B, 4] partial statistics, A [world, B] local argmax indices -> (lse, target logit, argmax) per row."""
    m = S[..., 0]
    M = m.max(0).values
    lse = M + torch.log((S[..., 1] * torch.exp(m - M)).sum(0))
    zt = (S[..., 2] * S[..., 3]).sum(0)
    r = m.argmax(0)                       # first maximum: the lowest rank = the lowest class index on ties
    return lse, zt, A.gather(0, r.unsqueeze(0))[0]


class _ShardedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, mod, label):
        B, N = x.shape[0], mod.world
        # TWO collectives in the forward (were four): the labels ride as one more fp32 column of the embeddings, the local argmax
        # as a fifth column of the row statistics -- class indices are < 2^24 and therefore exact in fp32 (checked in __init__)
        XY = mod._gather(torch.cat([x.contiguous().float(), label.contiguous().float().unsqueeze(1)], 1))      # [N, B, D + 1]
        X = XY[..., :-1].reshape(N * B, -1).contiguous()
        Y = XY[..., -1].reshape(N * B).to(torch.int64).contiguous()
        cos, st = mod._local_cosines(X)
        stats, arg = mod._partial_stats(cos, st, Y)
        SA = mod._gather(torch.cat([stats, arg.float().unsqueeze(1)], 1))                                       # [N, N*B, 5]
        lse, zt, amax = combine_row_stats(SA[..., :4], SA[..., 4].to(torch.int64))
        rows = slice(mod.rank * B, (mod.rank + 1) * B)
        ctx.mod, ctx.st, ctx.cos, ctx.Y, ctx.lse, ctx.B = mod, st, cos, Y, lse.contiguous(), B
        out_arg = amax[rows].contiguous()
        ctx.mark_non_differentiable(out_arg)
        return (lse - zt)[rows].mean(), out_arg

    @staticmethod
    def backward(ctx, dloss, _darg):
        mod, B = ctx.mod, ctx.B
        mod._check_gen(ctx.st)
        # every rank's rows are differentiated with THAT rank's upstream gradient of its local mean
        dl = mod._gather(dloss.reshape(1).float()).reshape(-1)
        row_scale = (dl / B).repeat_interleave(B).contiguous()
        dcos = mod._local_dcos(ctx.cos, ctx.st, ctx.Y, ctx.lse, row_scale)
        dX = mod._local_backward(ctx.st, dcos, ctx.cos)          # this shard's share of dX for all rows; dW accumulated in place
        return mod._reduce_scatter_rows(dX, B), None, None, None
