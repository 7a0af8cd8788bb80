"""Fused AdamW over flat parameter buffers + the linear schedule the reference's train script uses.

``FusedAdamW`` is a ``torch.optim.Optimizer`` (so ``transformers.get_scheduler`` / ``LambdaLR`` drive its
``param_groups[...]['lr']`` exactly as they drive torch's AdamW in multimodal_classifier_train.py:152-164), but
``step()`` is ONE HIP launch per flat buffer (mmsim_adamw_step) that also refreshes the bf16 shadow weights.
Semantics = torch.optim.AdamW defaults (betas 0.9/0.999, eps 1e-8, weight_decay 0.01).
Parameters outside any flat buffer (the reference's dead layers, which never receive gradients) are skipped, as
torch skips parameters whose ``.grad`` is None.
"""
import torch

from . import ops


def collect_flat_buffers(modules, dp_only=False):
    """Flat parameter buffers under the modules.  dp_only: leave out modules whose gradient is NOT exchanged between data-parallel
    ranks (``dp_exchange = False``: the class-sharded head, whose shard gradient is complete on its owner)."""
    if isinstance(modules, torch.nn.Module):
        modules = [modules]
    seen, out = set(), []
    for root in modules:
        for m in root.modules():
            fb = getattr(m, "flat_buffers", None)
            if fb is None or (dp_only and not getattr(m, "dp_exchange", True)):
                continue
            for f in fb():
                if id(f) not in seen:
                    seen.add(id(f))
                    out.append(f)
    return out


def linear_schedule_lr(lr0, t, warmup, total):
    """transformers.get_scheduler('linear') lambda (warmup may be a float, as in the reference: 0.15*T)."""
    if t < warmup:
        return lr0 * float(t) / float(max(1, warmup))
    return lr0 * max(0.0, float(total - t) / float(max(1, total - warmup)))


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, modules, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, grad_scale=1.0, exclude=()):
        skip = {id(f) for f in collect_flat_buffers(list(exclude))} if exclude else set()
        self.flats = [f for f in collect_flat_buffers(modules) if id(f) not in skip]
        if not self.flats:
            raise ValueError("FusedAdamW: no flat parameter buffers found under the given modules")
        if isinstance(modules, torch.nn.Module):
            modules = [modules]
        params, seen = [], {id(p) for e in exclude for p in e.parameters()}
        for m in modules:
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    params.append(p)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        # modules whose flat buffer is ONE row-normalised weight matrix (ArcMarginProduct): their update also leaves w_hat
        self._row_owners = {}
        for root in modules:
            for m in root.modules():
                if hasattr(m, "adamw_row_buffers") and hasattr(m, "flat_buffers"):
                    for f in m.flat_buffers():
                        self._row_owners[id(f)] = m
        self.grad_scale = grad_scale
        self._t = 0
        self._mv = {}
        # device float[3] {lr, 1 - beta1^t, 1 / sqrt(1 - beta2^t)} read by the kernels instead of the host values when set: the
        # step-dependent scalars of an update captured in a hipGraph (train.GraphedTrainStep writes it before every replay)
        self.dev_hyper = None

    def zero_grad(self, set_to_none=False, lazy=False):
        for f in self.flats:
            f.zero_grad(lazy=lazy)

    def _moments(self, f):
        mv = self._mv.get(id(f))
        if mv is None or mv[0].device != f.master.device:
            mv = (torch.zeros_like(f.master), torch.zeros_like(f.master))
            self._mv[id(f)] = mv
        return mv

    def _update(self, f, lo, hi):
        """The AdamW launch for elements [lo, hi) of flat buffer f (step count self._t), on the current stream."""
        g = self.param_groups[0]
        mv = self._moments(f)
        rn = self._row_owners.get(id(f))
        rows = rn.adamw_row_buffers() if rn is not None else None
        if rows is not None:
            # a row-normalised weight matrix (the ArcFace head): the update also leaves F.normalize(weight) for the next forward
            if lo != 0 or hi != f.total:
                raise ValueError("FusedAdamW: a row-normalised buffer is updated as a whole")
            R, D, w_hat, inv_norm = rows
            n = R * D
            ops.adamw_rows_l2norm(f.master[:n].view(R, D), f.grad[:n].view(R, D), mv[0][:n].view(R, D), mv[1][:n].view(R, D),
                                  w_hat, inv_norm, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._t,
                                  self.grad_scale, dev_hyper=self.dev_hyper)
            rn.mark_normalised()
            return
        ops.adamw_step(f.master[lo:hi], f.grad[lo:hi], mv[0][lo:hi], mv[1][lo:hi], f.shadow[lo:hi], g["lr"], g["betas"][0], g["betas"][1],
                       g["eps"], g["weight_decay"], self._t, self.grad_scale, dev_hyper=self.dev_hyper,
                       shadow16=None if f.shadow16 is None else f.shadow16[lo:hi])
        if rn is not None:
            rn.invalidate_normalised()          # the master changed under the module: its cached w_hat is stale

    @torch.no_grad()
    def step(self, closure=None):
        self._t += 1
        self._finish({})
        return None

    def _finish(self, done):
        """Update every element of every buffer that ``done`` ({id(flat): [(lo, hi), ...]}) does not cover."""
        if self.flats[0].master.is_cuda:
            ops.join_side_streams()      # gradients written by a tower on its own stream (multimodal_classifier.py)
        for f in self.flats:
            if f.grad is None:
                continue          # never produced a gradient: skipped like torch skips grad-is-None params
            f.ensure_device_state()
            f.materialize_zero()          # lazily zeroed and never written since: the update must see zeros
            pos = 0
            for lo, hi in sorted(done.get(id(f), [])) + [(f.total, f.total)]:
                if lo > pos:
                    self._update(f, pos, lo)
                pos = max(pos, hi)
            f._shadow_version = f.master._version

    # ---- the same step issued RANGE BY RANGE while the backward is still running (train.TrainStep, single process): the towers
    # report gradient ranges as they become final (grad_ready_hook), each range's update is launched at once on the caller's
    # optimiser stream, finish_ranged_step() does whatever was not reported.  Element for element the arithmetic of step().
    @torch.no_grad()
    def begin_ranged_step(self):
        self._t += 1
        self._done = {}

    @torch.no_grad()
    def step_range(self, f, lo, hi):
        if f.grad is None or hi <= lo:
            return
        f.ensure_device_state()
        self._update(f, lo, hi)
        self._done.setdefault(id(f), []).append((lo, hi))

    @torch.no_grad()
    def finish_ranged_step(self):
        done, self._done = self._done, None
        self._finish(done)

    # ---- checkpointing (the reference saves optimiser dicts next to the model, cv_classifier_train_daodian.py:298-306)
    def state_dict(self):
        """param_groups (lr / betas / ... as the schedulers left them) + step count + the exp_avg / exp_avg_sq of every flat
        buffer, in buffer order (the order of ``collect_flat_buffers`` over the modules the optimiser was built on)."""
        d = super().state_dict()
        d["mmsim_step"] = self._t
        d["mmsim_moments"] = [None if self._mv.get(id(f)) is None else tuple(t.detach().cpu().clone() for t in self._mv[id(f)])
                              for f in self.flats]
        return d

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        step, moments = sd.pop("mmsim_step", None), sd.pop("mmsim_moments", None)
        super().load_state_dict(sd)
        if step is None or moments is None:
            raise ValueError("FusedAdamW.load_state_dict: not a FusedAdamW state (mmsim_step / mmsim_moments missing)")
        if len(moments) != len(self.flats):
            raise ValueError(f"FusedAdamW.load_state_dict: {len(moments)} moment buffers for {len(self.flats)} flat buffers")
        self._t = int(step)
        self._mv = {}
        for f, mv in zip(self.flats, moments):
            if mv is None:
                continue
            if mv[0].numel() != f.master.numel():
                raise ValueError("FusedAdamW.load_state_dict: moment buffer size does not match the flat parameter buffer")
            self._mv[id(f)] = tuple(t.to(device=f.master.device, dtype=torch.float32).clone() for t in mv)


def hyper_values(opt, t):
    """{lr, 1 - beta1^t, 1 / sqrt(1 - beta2^t)} of step t (1-based) for ``opt``'s current learning rate -- computed exactly as
    mmsim_adamw_step computes them from its arguments (betas as the float32 values the C ABI receives, powers and the square root
    in double, the results rounded to float32), so that an update fed from device memory is bit-identical to one fed by value."""
    import math
    import struct
    f32 = lambda x: struct.unpack("f", struct.pack("f", x))[0]
    g = opt.param_groups[0]
    b1, b2 = f32(g["betas"][0]), f32(g["betas"][1])
    return [f32(g["lr"]), f32(1.0 - math.pow(b1, t)), f32(1.0 / math.sqrt(1.0 - math.pow(b2, t)))]


class FusedAdam(FusedAdamW):
    """``torch.optim.Adam`` with its defaults (betas 0.9 / 0.999, eps 1e-8, weight_decay 0) on the fused flat-buffer kernel --
    the optimiser of the reference's image-only loop (``torch.optim.Adam(model.parameters(), lr=1e-3)``,
    cv_classifier_train_daodian.py:264).  With weight_decay = 0 Adam and AdamW are the same update, so this is the AdamW launch
    with the decay term switched off; torch's coupled L2 form (weight_decay > 0 added to the gradient) is not implemented and
    is refused rather than silently replaced by the decoupled one."""

    def __init__(self, modules, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, exclude=()):
        if weight_decay != 0.0:
            raise ValueError("FusedAdam: coupled (L2) weight decay is not implemented; use FusedAdamW for decoupled decay")
        super().__init__(modules, lr=lr, betas=betas, eps=eps, weight_decay=0.0, grad_scale=grad_scale, exclude=exclude)
