"""fedmlp_amd: MI355X-native per-client training engine for the FedMLP hot path.

Host-side mirror of the reference call surface (build_model / LocalUpdate /
FedAvg*) over a C-ABI HIP library (include/fedmlp_hip.h).  Importing the
package does not touch the GPU; creating an engine without the built
``libfedmlp_hip.so`` raises (there is no CPU fallback).
"""
from . import spec  # noqa: F401

__all__ = ["spec"]
