"""diffusionvid_amd: MI355X-native DiffusionVID inference hot path.

Hand-written HIP/CDNA4 kernels behind a C ABI (csrc/, include/dvid_hip.h) and the host-side mirror
of the reference's plugin surface (mega_core.modeling detector / roi-head API, config keys,
BoxList results).  No CPU fallback: compute entry points raise if libdvid_hip.so or the GPU is
missing.
"""
__version__ = "0.1.0"
