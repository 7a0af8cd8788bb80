"""multimodalsimilar_amd -- MI355X-native (gfx950) compute path of the MultimodalSimilar two-tower + ArcFace
training step.  HIP kernels live in csrc/ behind the C ABI of include/mmsim_hip.h; the Python here is the
host side: device buffers (torch), stream plumbing, the layer schedules and the drop-in module API."""
from ._lib import lib, MmsimError  # noqa: F401

__all__ = ["lib", "MmsimError"]
