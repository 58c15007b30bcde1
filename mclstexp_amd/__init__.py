"""mclstexp_amd -- MI355X-native (gfx950) implementation of mclSTExp's contrastive training hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all hot-path
arithmetic outside the image backbone runs in hand-written HIP kernels behind the C-ABI declared
in ``include/mclstexp_hip.h`` (``libmclstexp_hip.so``).  There is no CPU fallback: calling an op
without the built library, or with non-GPU tensors, raises.
"""
__version__ = "0.1.0"
