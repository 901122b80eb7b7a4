"""dmvae_amd -- MI355X-native (gfx950) implementation of the DMVAE training hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed over RCCL);
the device work is hand-written HIP behind the C ABI in include/dmvae_hip.h, loaded by
``dmvae_amd._lib``.  There is no CPU fallback: calling an op without the built library or
without a GPU raises.
"""
__version__ = "0.1.0"
