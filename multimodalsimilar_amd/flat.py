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
    def __init__(self, specs, device="cpu"):
        """specs: ordered [(name, shape)].  Order is layout: neighbours are contiguous (q|k|v fusion)."""
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
        self._shadow_version = None

    def ensure_device_state(self):
        if self.grad is None or self.grad.device != self.master.device:
            self.grad = torch.zeros_like(self.master)
        if self.shadow is None or self.shadow.device != self.master.device:
            self.shadow = torch.empty(self.total, dtype=torch.bfloat16, device=self.master.device)
            self._shadow_version = None

    def sync_shadow(self, force=False):
        """Refresh the bf16 copies if torch-side code modified the master (the AdamW kernel keeps them in sync itself)."""
        self.ensure_device_state()
        v = self.master._version
        if force or self._shadow_version != v:
            ops.cast_to_bf16(self.master, self.shadow)
            self._shadow_version = v

    def zero_grad(self):
        if self.grad is not None:
            self.grad.zero_()
