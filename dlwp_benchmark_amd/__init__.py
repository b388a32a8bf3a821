"""MI355X-native autoregressive rollout engine for the dlwpbench backbones.

Host side: the reference's model registry / constructor kwargs / forward signature
(`dlwp_benchmark_amd.models`), device side: libdlwp_hip.so (hand-written gfx950 kernels behind
the C ABI of include/dlwp_hip.h).  See DESIGN.md.
"""
__version__ = "0.1.0"
