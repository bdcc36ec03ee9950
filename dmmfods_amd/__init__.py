"""dmmfods_amd -- MI355X (gfx950) implementation of the DMMFODS Dense_U_Net_lidar training hot path.

Package layout mirrors the reference's Pytorch-Project-Template directories (README.md:27-29):
``graphs/models``, ``graphs/losses``, ``agents``, ``utils``; compute lives in ``csrc`` (HIP kernels behind the
C ABI declared in ``include/dmmfods_hip.h``), reached through ``_lib`` (ctypes)."""
__version__ = "0.1.0"
