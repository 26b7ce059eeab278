"""shmgan_amd -- MI355X (gfx950) native SHMGAN generator+discriminator train_step.

Host side in Python (like the reference), compute in hand-written HIP kernels reached through
the C ABI of libshmgan_hip.so (include/shmgan_hip.h).  There is no CPU / PyTorch fallback.
"""
from .trainer import LOSS_NAMES, KernelAbortError, ShmGANwithSSpecSeg  # noqa: F401
from .model import Discriminator, Generator  # noqa: F401

__all__ = ["ShmGANwithSSpecSeg", "Generator", "Discriminator", "LOSS_NAMES"]
