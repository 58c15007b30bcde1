"""mclstexp_amd -- MI355X-native (gfx950) implementation of mclSTExp's contrastive training hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all hot-path
arithmetic outside the image backbone runs in hand-written HIP kernels behind the C-ABI declared
in ``include/mclstexp_hip.h`` (``libmclstexp_hip.so``).  There is no CPU fallback: calling an op
without the built library, or with non-GPU tensors, raises.
"""
import os as _os

# Kernel arguments in device memory: 1 ms of a 13.4 ms training step on MI355X (bench.py, A/B).  The HIP runtime reads the
# flag when its library is loaded, so this only takes effect when the package is imported before torch.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
