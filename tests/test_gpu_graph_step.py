"""The training step replayed as one hipGraph (train.GraphedTrainStep) against the eager TrainStep: same kernels, so in
deterministic mode the losses and parameters are BIT-IDENTICAL step by step -- including the values a captured graph would
freeze (learning-rate schedules with warm-up, Adam bias corrections), which the kernels read from device memory; dropout draws
new masks on every replay through the device step word."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def deterministic():
    from multimodalsimilar_amd import ops
    ops.set_deterministic(True)
    yield
    ops.set_deterministic(False)


def _model(cfg, dropout):
    from multimodalsimilar_amd import train as T
    return T.build_model(cfg, "cuda", seed=0, dropout=dropout)


@pytest.mark.parametrize("name", ["tiny", "nlp"])
def test_graph_replay_equals_eager_steps_bit_for_bit(name, deterministic):
    from multimodalsimilar_amd import train as T
    cfg = dict(T.CONFIGS["tiny"]) if name == "tiny" else dict(kind="nlp", text="tiny", seq_len=32, batch=16, classes=64)
    steps, total = 7, 12                       # warm-up of the head schedule is 0.15 * 12 = 1.8 steps: the lr changes every step
    batches = [T.synthetic_batch(cfg, "cuda", seed=30 + i) for i in range(steps)]
    m1 = _model(cfg, False)
    ts1 = T.TrainStep(m1, cfg["kind"], total)
    eager = [ts1.step(b)[0].item() for b in batches]
    m2 = _model(cfg, False)
    ts2 = T.TrainStep(m2, cfg["kind"], total)
    # GraphedTrainStep warms up with 2 REAL steps on the batch it is given, then captures: feed it the same sequence
    seq = iter(batches)
    first = next(seq)
    g = T.GraphedTrainStep.__new__(T.GraphedTrainStep)
    # warm-up steps must see batches 0 and 1: drive the pieces by hand (the constructor uses one batch for both)
    from multimodalsimilar_amd.optim import hyper_values
    from multimodalsimilar_amd._lib import lib
    g.ts, g._hv, g._lib = ts2, hyper_values, lib
    g.static = {k: v.clone() for k, v in first.items()}
    g.hyper = torch.zeros(8, device="cuda"); g.seed = torch.zeros(1, dtype=torch.int64, device="cuda")
    g._init_host_ring()
    g.replays = 0
    ts2.opt_emb.dev_hyper, ts2.opt_fc.dev_hyper = g.hyper[0:3], g.hyper[4:7]
    lib.set_step_seed_ptr(g.seed.data_ptr())
    try:
        graphed = []
        for b in batches[:2]:                  # eager steps WITH the device-side scalars (what the warm-up runs)
            for k, v in b.items():
                g.static[k].copy_(v)
            g._prepare()
            l, _ = ts2._body(g.static)
            ts2._advance()
            graphed.append(l.item())
        g._capture(warmup=0)
        for b in batches[2:]:
            l, pred = g.step(b)
            graphed.append(l.item())
        assert graphed == eager, (graphed, eager)
        sd1, sd2 = m1.state_dict(), m2.state_dict()
        diff = [k for k in sd1 if not torch.equal(sd1[k], sd2[k])]
        assert not diff, diff[:6]
        assert ts1.t == ts2.t == steps and ts1.opt_fc._t == ts2.opt_fc._t == steps
        assert abs(ts1.opt_fc.param_groups[0]["lr"] - ts2.opt_fc.param_groups[0]["lr"]) < 1e-15
    finally:
        g.close()


def test_graph_replay_draws_new_dropout_masks_every_step(deterministic):
    # deterministic reductions: without dropout the frozen model's loss must not move AT ALL between replays (with the default
    # atomics it moves by ~1e-3 from run to run, which is the size of the effect this test separates from)
    from multimodalsimilar_amd import train as T
    cfg = dict(T.CONFIGS["tiny"])
    batch = T.synthetic_batch(cfg, "cuda", seed=3)
    losses = {}
    for dropout in (False, True):
        model = _model(cfg, dropout)
        ts = T.TrainStep(model, cfg["kind"], 10 ** 9, lr_emb=0.0, lr_fc=0.0)       # frozen weights: only the masks can move the loss
        g = T.GraphedTrainStep(ts, batch, warmup=2)
        try:
            losses[dropout] = [g.step(batch)[0].item() for _ in range(4)]
        finally:
            g.close()
    assert max(losses[False]) - min(losses[False]) < 1e-3 * abs(losses[False][0])                   # BatchNorm running stats only
    assert len(set(losses[True])) == 4 and max(losses[True]) - min(losses[True]) > 1e-3             # a new mask per replay


def test_capture_starts_from_a_canonical_gradient_state(deterministic):
    """The head's gradient buffer is zeroed lazily by the step (flat.zero_grad(lazy=True): the dW product overwrites).  A capture
    taken while the buffer is NOT marked (really zero: e.g. right after a checkpoint load or a manual zero_grad) must not record
    an accumulating dW product -- replays would then sum the gradients of all steps."""
    from multimodalsimilar_amd import train as T
    cfg = dict(kind="nlp", text="tiny", seq_len=32, batch=16, classes=64)
    steps, total = 6, 12
    batches = [T.synthetic_batch(cfg, "cuda", seed=50 + i) for i in range(steps)]
    m1 = _model(cfg, False)
    ts1 = T.TrainStep(m1, cfg["kind"], total)
    eager = [ts1.step(b)[0].item() for b in batches]
    m2 = _model(cfg, False)
    ts2 = T.TrainStep(m2, cfg["kind"], total)
    got = [ts2.step(b)[0].item() for b in batches[:2]]
    for f in ts2.opt_fc.flats:
        assert f.zero_pending                       # the step left the head's buffer marked, not filled
        f.materialize_zero()                        # ... now it is really zero and unmarked
    g = T.GraphedTrainStep(ts2, batches[2], warmup=0)
    try:
        got += [g.step(b)[0].item() for b in batches[2:]]
    finally:
        g.close()
    assert max(abs(a - b) for a, b in zip(got, eager)) < 1e-5 * abs(eager[0]), (got, eager)


def test_optimiser_updates_issued_during_the_backward_are_bit_identical(deterministic, monkeypatch):
    """MMSIM_OPT_IN_BWD=1: the head's and each encoder layer's AdamW range launched from the towers' grad_ready hooks on a side
    stream (optim.FusedAdamW.step_range) against the two updates at the end of the step -- same kernels on the same data, so the
    losses and every parameter are bit-identical; every element is updated exactly once (the unreported ranges at the end)."""
    from multimodalsimilar_amd import train as T
    cfg = dict(T.CONFIGS["tiny"])
    batches = [T.synthetic_batch(cfg, "cuda", seed=70 + i) for i in range(4)]
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MMSIM_OPT_IN_BWD", flag)
        m = _model(cfg, False)
        ts = T.TrainStep(m, cfg["kind"], 10)
        assert (ts._oib is not None) == (flag == "1")
        losses = [ts.step(b)[0].item() for b in batches]
        torch.cuda.synchronize()
        out.append((losses, {k: v.clone() for k, v in m.state_dict().items()}))
        if flag == "1":
            done = ts.opt_emb._done
            assert done is None                     # finish_ranged_step consumed the ranges
    (l0, sd0), (l1, sd1) = out
    assert l0 == l1, (l0, l1)
    diff = [k for k in sd0 if not torch.equal(sd0[k], sd1[k])]
    assert not diff, diff[:6]
