"""Class-sharded ArcFace head on the HIP kernels (two ranks sharing the one GPU of the box, gloo rendezvous -- RCCL refuses
duplicate devices; with >= 2 visible devices the same test also runs over RCCL, one process per GPU): loss, argmax, dX and the
shard's dW against the replicated HIP head on the same global batch, and one TrainStep with the sharded head against one with
the replicated head."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return torch.device("cuda", dev)


def _head_worker(rank, world, port, backend, out):
    dev = _setup(rank, world, port, backend)
    from multimodalsimilar_amd.head import ArcMarginProduct
    from multimodalsimilar_amd.sharded_head import ShardedArcMarginProduct
    B, D, C, m = 16, 64, 1003, 0.5                   # ragged shards: 502 + 501
    g = torch.Generator().manual_seed(3)
    X = torch.randn(world * B, D, generator=g)
    W = torch.randn(C, D, generator=g) * 0.2
    Y = torch.randint(0, C, (world * B,), generator=g)
    Y[1], Y[B + 1] = 501, 502                        # targets on both sides of the shard boundary
    sh = ShardedArcMarginProduct(D, C, s=64.0, m=m, full_weight=W).to(dev)
    x = X[rank * B:(rank + 1) * B].to(dev).requires_grad_(True)
    y = Y[rank * B:(rank + 1) * B].to(dev)
    up = 1.0 + 0.5 * rank
    loss, arg = sh.forward_loss(x, y)
    (loss * up).backward()
    sh.check_labels()
    # replicated HIP head, this rank's rows
    rep = ArcMarginProduct(D, C, s=64.0, m=m)
    with torch.no_grad():
        rep._flat.view("weight").copy_(W)
    rep.to(dev)
    xr = X[rank * B:(rank + 1) * B].to(dev).requires_grad_(True)
    lr, ar = rep.forward_loss(xr, y)
    (lr * up).backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - lr.item()) < 2e-3 * abs(lr.item()), (loss.item(), lr.item())
    assert torch.equal(arg, ar)
    rel = lambda a, b: ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
    assert rel(x.grad, xr.grad) < 1e-2
    gw = rep._flat.gview("weight").detach().cpu().clone()          # this rank's rows' contribution to the full dW
    dist.all_reduce(gw, op=dist.ReduceOp.SUM) if backend == "gloo" else None
    if backend == "nccl":
        gwd = gw.to(dev); dist.all_reduce(gwd, op=dist.ReduceOp.SUM); gw = gwd.cpu()
    c0, cl = sh.class_offset, sh.local_classes
    assert rel(sh._flat.gview("weight").detach().cpu(), gw[c0:c0 + cl]) < 1e-2
    v, i = sh.predict(x.detach())
    cosr = rep.forward_test(xr.detach())
    assert torch.equal(i, cosr.argmax(1)) and rel(v, cosr.max(1).values) < 1e-2
    if rank == 0:
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def _step_worker(rank, world, port, backend, out, select="force"):
    """select: how the run asks for the sharded head -- "force" (cfg["sharded_head"]) or "threshold" (cfg["shard_head_from"] <= classes,
    what bench.py --shard-head-from / MMSIM_SHARD_HEAD_FROM set: VERDICT r3 item 8, cfg4's 100 000-class head sharded)."""
    dev = _setup(rank, world, port, backend)
    from multimodalsimilar_amd import train as T
    cfg = dict(kind="nlp", text="tiny", seq_len=32, batch=8, classes=250)
    res = {}
    for sharded in (False, True):
        pick = dict(sharded_head=sharded) if select == "force" else dict(shard_head_from=200 if sharded else 10 ** 9)
        model = T.build_model(dict(cfg, **pick), dev, seed=0, dropout=False)
        assert (type(model.classifier).__name__ == "ShardedArcMarginProduct") == sharded
        ts = T.TrainStep(model, "nlp", 10)
        losses = []
        for i in range(3):
            full = T.synthetic_batch(dict(cfg, batch=16), dev, seed=40 + i)
            half = {k: v[rank * 8:(rank + 1) * 8] for k, v in full.items()}
            l, pred = ts.step(half)
            losses.append(l.item())
        torch.cuda.synchronize()
        model.classifier.check_labels()
        w = model.classifier.weight.detach().cpu()
        if sharded:
            c0, cl = model.classifier.class_offset, model.classifier.local_classes
            w_full_ref = res["w"]
            d = (w - w_full_ref[c0:c0 + cl]).abs().mean().item()
            upd = (w_full_ref[c0:c0 + cl] - res["w0"][c0:c0 + cl]).abs().mean().item()
            assert d < 0.1 * upd + 1e-9, (d, upd)
            for a, b in zip(losses, res["losses"]):
                assert abs(a - b) < 5e-3 * abs(b), (losses, res["losses"])
            k = "ptm.encoder.layer.1.output.dense.weight"
            dt = (model.state_dict()[k].cpu() - res["tower"]).abs().mean().item()
            assert dt < 2e-5, dt                       # the towers see the same dX
        else:
            res = dict(w=w, losses=losses, tower=model.state_dict()["ptm.encoder.layer.1.output.dense.weight"].cpu().clone())
            torch.manual_seed(0)
            res["w0"] = T.build_model(dict(cfg, sharded_head=False), "cpu", seed=0, dropout=False).classifier.weight.detach().clone()
    if rank == 0:
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def _backends():
    return ["gloo"] + (["nccl"] if torch.cuda.device_count() >= 2 else [])


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_sharded_head_matches_the_replicated_head(tmp_path, backend):
    if backend not in _backends():
        pytest.skip("RCCL needs one GPU per rank: fewer than 2 devices visible")
    out = str(tmp_path / "ok.txt")
    mp.spawn(_head_worker, args=(2, _free_port(), backend, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


@pytest.mark.parametrize("select", ["force", "threshold"])
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_train_step_with_the_sharded_head_tracks_the_replicated_one(tmp_path, backend, select):
    if backend not in _backends():
        pytest.skip("RCCL needs one GPU per rank: fewer than 2 devices visible")
    out = str(tmp_path / "ok.txt")
    mp.spawn(_step_worker, args=(2, _free_port(), backend, out, select), nprocs=2, join=True)
    assert open(out).read() == "ok"
