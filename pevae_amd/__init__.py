"""pevae_amd -- MI355X-native kernels behind PEVAE's DLS_VAE training step.

The package mirrors the reference's module surface (models/enc2dec, models/nn_lib, models/criterions) so the
classes drop in behind main_re.py's model code; all arithmetic runs in hand-written HIP kernels from
pevae_amd/lib/libpevae_hip.so (C ABI: include/pevae_hip.h).  No CPU fallback exists.
"""
from . import _lib

__all__ = ["_lib"]
__version__ = "0.1.0"
