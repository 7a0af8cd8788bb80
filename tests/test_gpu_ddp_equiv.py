"""Data-parallel equivalence on the GPU box: two ranks (two processes sharing the one GPU, gloo rendezvous on
127.0.0.1 -- RCCL refuses duplicate devices, the exchange code path is otherwise the same) each take half of a batch;
after two steps their parameters must equal a single process stepping on the whole batch: the update is the gradient of
the GLOBAL-batch mean loss (reference semantics, nlp_classifier_train_daodian_v2_dist.py:139-144)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
CFG = dict(kind="nlp", text="tiny", seq_len=32, batch=16, classes=64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, backend):
    dev = rank if backend == "nccl" else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(dev),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from multimodalsimilar_amd import train as T
    torch.cuda.set_device(dev)
    if backend == "nccl":      # RCCL over xGMI, one process per GPU: the production exchange (async all-reduce of ranges written on
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))      # tower side streams)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    model = T.build_model(CFG, "cuda", seed=0, dropout=False)
    ts = T.TrainStep(model, "nlp", 10)
    assert ts.exchange is not None and ts.exchange.world == 2
    for i in range(2):      # two steps: the head's first warm-up step runs at lr 0 (reference order), the second one moves it
        full = T.synthetic_batch(CFG, "cuda", seed=5 + i)
        half = {k: v[rank * 8:(rank + 1) * 8] for k, v in full.items()}
        loss, _ = ts.step(half)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_ranks_equal_one_process_on_the_whole_batch(tmp_path, backend):
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank: fewer than 2 devices visible on this box")
    from multimodalsimilar_amd import train as T
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, backend), nprocs=2, join=True)
    ddp = torch.load(out)
    model = T.build_model(CFG, "cuda", seed=0, dropout=False)
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ts = T.TrainStep(model, "nlp", 10)
    for i in range(2):
        ts.step(T.synthetic_batch(CFG, "cuda", seed=5 + i))
    torch.cuda.synchronize()
    single = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    checked = 0
    for k in ("classifier.weight", "ptm.encoder.layer.1.output.dense.weight", "ptm.encoder.layer.0.attention.self.query.weight",
              "ptm.embeddings.word_embeddings.weight", "ptm.pooler.dense.bias"):
        upd = (single[k] - sd0[k]).abs().mean().item()
        diff = (single[k] - ddp[k]).abs().mean().item()
        assert upd > 0, k                                  # the step moved the parameter ...
        assert diff < 0.05 * upd + 1e-9, (k, diff, upd)     # ... and both ways of running it moved it the same way
        checked += 1
    assert checked == 5


def _worker_rccl_one_rank(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from multimodalsimilar_amd import train as T
    from multimodalsimilar_amd.dist import GradientExchange
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg = dict(T.CONFIGS["tiny"])
    model = T.build_model(cfg, "cuda", seed=0, dropout=False)
    ts = T.TrainStep(model, cfg["kind"], 10)
    assert ts.exchange is None                       # one rank: TrainStep does not exchange ...
    ts.exchange = GradientExchange(model, bucket_bytes=1 << 16, always_exchange=True)      # ... unless told to; small buckets: many collectives
    n_coll = [0]
    launch = ts.exchange._launch
    def counting(flat, s, e):
        n_coll[0] += 1
        return launch(flat, s, e)
    ts.exchange._launch = counting
    for i in range(3):
        ts.step(T.synthetic_batch(cfg, "cuda", seed=5 + i))
    torch.cuda.synchronize()
    assert dist.get_backend() == "nccl" and n_coll[0] >= 6, n_coll
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, out)
    dist.destroy_process_group()


def test_rccl_collectives_execute_on_one_rank(tmp_path, monkeypatch):
    """RCCL itself on the one GPU of this box: a one-rank "nccl" group with the exchange forced on.  Both towers' gradient ranges
    (the image tower's are written on its side stream) go through dist.all_reduce(async_op=True) on RCCL's stream and finish();
    a sum over one rank is the identity, so three steps must equal the plain single-process steps exactly."""
    from multimodalsimilar_amd import ops
    from multimodalsimilar_amd import train as T
    out = str(tmp_path / "rccl1.pt")
    monkeypatch.setenv("MMSIM_DETERMINISTIC", "1")            # inherited by the worker: both runs take the reproducible kernels
    mp.spawn(_worker_rccl_one_rank, args=(_free_port(), out), nprocs=1, join=True)
    got = torch.load(out)
    ops.set_deterministic(True)
    try:
        cfg = dict(T.CONFIGS["tiny"])
        model = T.build_model(cfg, "cuda", seed=0, dropout=False)
        ts = T.TrainStep(model, cfg["kind"], 10)
        for i in range(3):
            ts.step(T.synthetic_batch(cfg, "cuda", seed=5 + i))
        torch.cuda.synchronize()
        ref = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    finally:
        ops.set_deterministic(False)
    bad = [k for k in ref if not torch.equal(ref[k], got[k])]
    assert not bad, bad[:5]
