"""keynet_amd -- MI355X-native keyed-forward engine behind the visym/keynet API surface.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); all arithmetic of the keyed
forward runs in hand-written gfx950 HIP kernels behind the C ABI declared in include/keynet_hip.h.
"""
__version__ = '0.1.0'
