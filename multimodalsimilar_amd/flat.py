"""Flat parameter storage: one fp32 master buffer, one fp32 gradient buffer and one bf16 shadow buffer per
tower, with the module's nn.Parameters as views.

Why flat: the fused AdamW step is ONE launch over the whole tower, the data-parallel gradient exchange is a
handful of large contiguous RCCL all-reduces (xGMI is per-link bound: few big messages), zero_grad is one
memset, and the bf16 copies the MFMA kernels read are refreshed by the optimiser kernel itself.
"""
import torch

from . import ops

ALIGN = 8   # elements: every tensor starts 16-byte aligned in the bf16 shadow (32-byte in fp32)


class FlatBuffer:
    def __init__(self, specs, device="cpu", f16_shadow=False):
        """specs: ordered [(name, shape)].  Order is layout: neighbours are contiguous (q|k|v fusion).
        f16_shadow: also keep an fp16 copy (``shadow16`` / ``sview16``) -- the image tower's forward products read fp16 weights
        (fp16 activations, csrc/common.h), its data-gradient products the bf16 copy."""
        self.f16_shadow = f16_shadow
        self.shadow16 = None
        self.names = [n for n, _ in specs]
        self.shapes = {n: tuple(s) for n, s in specs}
        self.offsets = {}
        off = 0
        for n, s in specs:
            self.offsets[n] = off
            numel = 1
            for d in s:
                numel *= d
            off += ops.round_up(numel, ALIGN)
        self.total = ops.round_up(off, ALIGN)
        self.master = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad = None
        self.shadow = None
        self._shadow_version = None

    # ---- views
    def _view(self, buf, name, shape=None):
        o = self.offsets[name]
        shape = self.shapes[name] if shape is None else shape
        n = 1
        for d in shape:
            n *= d
        return buf[o:o + n].view(shape)

    def view(self, name):
        return self._view(self.master, name)

    def gview(self, name, shape=None):
        return self._view(self.grad, name, shape)

    def sview(self, name, shape=None):
        return self._view(self.shadow, name, shape)

    def sview16(self, name, shape=None):
        return self._view(self.shadow16, name, shape)

    def span(self, first, last):
        """[start, end) element range covering tensors first..last (inclusive, in layout order)."""
        o = self.offsets[last]
        n = 1
        for d in self.shapes[last]:
            n *= d
        return self.offsets[first], ops.round_up(o + n, ALIGN)

    # ---- device state
    def apply_(self, fn):
        self.master = fn(self.master)
        self.grad = None
        self.shadow = None
        self.shadow16 = None
        self._shadow_version = None

    def ensure_device_state(self):
        if self.grad is None or self.grad.device != self.master.device:
            self.grad = torch.zeros_like(self.master)
        if self.shadow is None or self.shadow.device != self.master.device:
            self.shadow = torch.empty(self.total, dtype=torch.bfloat16, device=self.master.device)
            self._shadow_version = None
        if self.f16_shadow and (self.shadow16 is None or self.shadow16.device != self.master.device):
            self.shadow16 = torch.empty(self.total, dtype=torch.float16, device=self.master.device)
            self._shadow_version = None

    def sync_shadow(self, force=False):
        """Refresh the bf16 copies if torch-side code modified the master (the AdamW kernel keeps them in sync itself)."""
        self.ensure_device_state()
        v = self.master._version
        if force or self._shadow_version != v:
            ops.cast_to_bf16(self.master, self.shadow)
            if self.f16_shadow:
                ops.cast_to_f16(self.master, self.shadow16)
            self._shadow_version = v

    def zero_grad(self, lazy=False):
        """Zero the gradient buffer.  lazy=True (the training step's own zero_grad of a buffer whose ONLY writer can overwrite,
        see take_zero_pending) only marks it: the next writer overwrites instead of accumulating and the 1.1 GB fill + read of
        the ArcFace head's gradient never happen; any other reader calls materialize_zero() first."""
        if self.grad is None:
            return
        if lazy and self.overwrite_capable:
            self.zero_pending = True
            return
        self.zero_pending = False
        self.grad.zero_()

    # a lazily zeroed buffer: True between zero_grad(lazy=True) and the first writer / reader
    zero_pending = False
    overwrite_capable = False          # set by the owning module when one full-buffer writer produces the whole gradient

    def take_zero_pending(self):
        """Called by the (single, full-coverage) writer: True = write with overwrite semantics, the buffer counts as zeroed."""
        p, self.zero_pending = self.zero_pending, False
        return p

    def materialize_zero(self):
        if self.zero_pending and self.grad is not None:
            self.grad.zero_()
        self.zero_pending = False
