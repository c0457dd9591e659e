"""halo2-vectordb proving hot path, MI355X-native (gfx950).

Layout:
  csrc/        hand-written HIP kernels + the C ABI (include/vdb.h) -> libvdb_hip.so
  host/        C++ mirror of the reference's Rust gadget interface above the C ABI
  api.py       numpy/ctypes plumbing used by tests and bench.py
No CPU fallback exists in this package; the oracle under /oracle is test infrastructure only.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
