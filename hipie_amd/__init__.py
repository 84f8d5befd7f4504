"""hipie_amd -- MI355X-native (gfx950) implementation of HIPIE's single-image inference hot path.

Python host code on PyTorch-ROCm (device memory, streams, library GEMMs, torch.distributed) calling the hand-written
HIP kernels of hipie_amd/csrc through the C ABI declared in include/hipie_mi355.h.  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"
