"""sobfu_amd -- MI355X-native SobolevFusion solver hot path (HIP kernels for gfx950 behind a C ABI).

Python here is plumbing (device memory via torch, streams, torch.distributed); the product is
libsobfu_hip.so (sobfu_amd/csrc, include/sobfu_hip.h).
"""
__version__ = "0.1.0"
