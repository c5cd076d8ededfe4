"""pylrbms_amd -- MI355X-native hot path of dune-community/pylrbms.

Host side (this package) mirrors the reference's Python API for the path; all arithmetic runs in
hand-written HIP kernels (pylrbms_amd/csrc) behind the C ABI declared in include/lrbms_hip.h.
"""
__version__ = '0.1.0'
