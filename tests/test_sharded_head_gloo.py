"""Class-sharded ArcFace head, exchange logic on CPU: world_size-2 gloo run in which every rank owns half of the classes.
The local maths (cosines, margin, partial softmax statistics, gradients) comes from plain torch here -- the product class
runs those four steps on the HIP kernels (tests/test_gpu_sharded_head.py) -- so what this test pins is everything the
sharding adds: the gathers, the combination of the per-rank row statistics, which rank applies the margin, the per-rank
upstream gradients, the reduce-scatter and the gradient conventions (dX = d(local mean), a shard's dW = sum over the ranks'
local means).  Expected values: the oracle's replicated head on the concatenated batch."""
import math
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_cpu_sharded():
    from multimodalsimilar_amd.sharded_head import ShardedArcMarginProduct

    class CpuSharded(ShardedArcMarginProduct):
        """Test double: the four local steps in fp32 torch (autograd) instead of the HIP kernels."""

        def _check_gen(self, st):
            pass

        def _local_logits(self, X, W, Y):
            cos = F.normalize(X) @ F.normalize(W).t()                                  # arcface.py:47
            sine = torch.sqrt((1.0 - cos.pow(2)).clamp_min(1e-12))                     # :49
            phi = cos * self.cos_m - sine * self.sin_m                                 # :50
            phi = torch.where(cos > 0, phi, cos) if self.easy_margin else torch.where((cos - self.th) > 0, phi, cos - self.mm)
            yl = Y - self.class_offset
            here = (yl >= 0) & (yl < self.local_classes)
            onehot = torch.zeros_like(cos)
            onehot[here, yl[here]] = 1.0
            return self.s * (onehot * phi + (1.0 - onehot) * cos), onehot            # :58-61

        def _local_cosines(self, X):
            Xl = X.detach().clone().requires_grad_(True)
            Wl = self.weight.detach().clone().requires_grad_(True)
            return None, dict(B=X.shape[0], X=Xl, W=Wl, gen=0)

        def _partial_stats(self, cos, st, Y):
            with torch.enable_grad():             # autograd.Function.forward runs with grad mode off
                z, onehot = self._local_logits(st["X"], st["W"], Y)
            st["z"], st["onehot"] = z, onehot
            zd = z.detach()
            m, a = zd.max(1)
            stats = torch.stack([m, torch.exp(zd - m[:, None]).sum(1), (zd * onehot).sum(1), onehot.sum(1)], 1)
            return stats, a + self.class_offset

        def _local_dcos(self, cos, st, Y, lse, row_scale):
            return (torch.exp(st["z"].detach() - lse[:, None]) - st["onehot"]) * row_scale[:, None]      # dLoss/dlogits

        def _local_backward(self, st, dlogits, cos):
            st["z"].backward(dlogits)
            if self._flat.grad is None:
                self._flat.grad = torch.zeros_like(self._flat.master)
            self._flat.gview("weight").add_(st["W"].grad)
            return st["X"].grad

    return CpuSharded


def _data(world, B, D, C, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(world * B, D, generator=g)
    W = torch.randn(C, D, generator=g) * 0.3
    Y = torch.randint(0, C, (world * B,), generator=g)
    Y[0], Y[B] = 0, C - 1                   # both shards' edge classes occur
    return X, W, Y


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import arcface_ref
    B, D, C, m = 6, 16, 11, 0.5                      # 11 classes over 2 ranks: shards of 6 and 5
    X, W, Y = _data(world, B, D, C)
    head = _make_cpu_sharded()(D, C, s=64.0, m=m, full_weight=W)
    assert (head.class_offset, head.local_classes, head.total_classes) == ((0, 6) if rank == 0 else (6, 5)) + (C,)
    x = X[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    y = Y[rank * B:(rank + 1) * B]
    up = 1.0 + 0.5 * rank                             # a different upstream gradient per rank
    loss, arg = head.forward_loss(x, y)
    (loss * up).backward()
    # ---- oracle: the replicated head on the whole batch; per-rank local means
    Xo, Wo = X.clone().requires_grad_(True), W.clone().requires_grad_(True)
    logits = arcface_ref.arcface_forward(Xo, Wo, Y, 64.0, m)
    per_row = F.cross_entropy(logits, Y, reduction="none")
    total = sum((1.0 + 0.5 * r) * per_row[r * B:(r + 1) * B].mean() for r in range(world))
    total.backward()
    assert abs(loss.item() - per_row[rank * B:(rank + 1) * B].mean().item()) < 1e-4 * abs(loss.item())
    assert torch.equal(arg, logits.argmax(1)[rank * B:(rank + 1) * B])
    assert torch.allclose(x.grad, Xo.grad[rank * B:(rank + 1) * B], rtol=1e-4, atol=1e-6)
    c0, cl = head.class_offset, head.local_classes
    assert torch.allclose(head._flat.gview("weight"), Wo.grad[c0:c0 + cl], rtol=1e-4, atol=1e-6)
    # evaluation paths: all-gathered cosines and the argmax exchange
    class _Eval(type(head)):
        def _local_cosines(self, Xg):
            return F.normalize(Xg) @ F.normalize(self.weight.detach()).t(), None
    head.__class__ = _Eval
    cos_full = F.normalize(X) @ F.normalize(W).t()
    got = head.forward_test(x.detach())
    assert got.shape == (B, C) and torch.allclose(got, cos_full[rank * B:(rank + 1) * B], atol=1e-6)
    v, i = head.predict(x.detach())
    assert torch.equal(i, cos_full[rank * B:(rank + 1) * B].argmax(1)) and torch.allclose(v, cos_full[rank * B:(rank + 1) * B].max(1).values, atol=1e-6)
    # the gradient exchange must leave the shard alone
    from multimodalsimilar_amd.dist import GradientExchange
    root = torch.nn.Module()
    root.classifier = head
    ex = GradientExchange(root)
    assert ex.flats == [] and head.grad_ready_hook is None
    if rank == 0:
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_head_world2_equals_the_replicated_head(tmp_path):
    out = str(tmp_path / "ok.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_shard_ranges_cover_the_classes_exactly():
    from multimodalsimilar_amd.sharded_head import shard_range, combine_row_stats
    for C, N in ((11, 2), (1000000, 8), (100000, 3), (8, 8), (9, 8)):
        spans = [shard_range(C, N, r) for r in range(N)]
        assert spans[0][0] == 0 and sum(n for _, n in spans) == C
        for (a, n), (b, _) in zip(spans, spans[1:]):
            assert a + n == b or n == 0
    # statistics of one row split over two ranks combine to the row's log-sum-exp, and ties take the lower class
    z = torch.tensor([[1.0, 3.0, 3.0, -2.0]])
    S = torch.stack([torch.tensor([[3.0, math.exp(-2.0) + 1.0, 0.0, 0.0]]), torch.tensor([[3.0, 1.0 + math.exp(-5.0), 3.0, 1.0]])])
    lse, zt, am = combine_row_stats(S, torch.tensor([[1], [2]]))
    assert torch.allclose(lse, torch.logsumexp(z, 1)) and zt.item() == 3.0 and am.item() == 1


def _threshold_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodalsimilar_amd import train as T
    cfg4 = dict(T.CONFIGS["cfg4"])
    assert not T._want_sharded_head(cfg4)                                        # 100 000 classes: below the default threshold
    assert T._want_sharded_head(dict(cfg4, shard_head_from=100000))             # bench.py --shard-head-from 100000
    assert not T._want_sharded_head(dict(cfg4, shard_head_from=100000, sharded_head=False))      # the explicit switch wins
    assert T._want_sharded_head(dict(T.CONFIGS["cfg5"]))                         # 1 M classes: sharded by default
    assert T._want_sharded_head(dict(cfg4, sharded_head=True))
    if rank == 0:
        open(out, "w").write("ok")
    dist.destroy_process_group()


def test_shard_threshold_selects_the_head_under_data_parallelism(tmp_path):
    out = str(tmp_path / "ok.txt")
    mp.spawn(_threshold_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"
    from multimodalsimilar_amd import train as T
    assert not T._want_sharded_head(dict(T.CONFIGS["cfg5"]))                     # single process: never sharded
