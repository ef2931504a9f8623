"""charls_amd -- MI355X-native JPEG-LS scan engine behind the CharLS C ABI.

Python is plumbing here: `capi` binds the C ABI (ctypes) for tests and benchmarks, `batch` feeds device-resident
frames (torch tensors are only used for HBM allocation, streams and torch.distributed), `synth` makes seeded frames.
The product is charls_amd/lib/libcharls_amd.so (charls_amd/csrc, include/charls_amd.h).
"""
__all__ = ["capi", "batch", "synth", "build"]
