"""ctypes loader for libvdb_hip.so — the C-ABI shared library (include/vdb.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C halo2_vectordb_amd/csrc`.
There is no CPU fallback: if the library is missing or no GPU is visible, calls raise VdbError.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvdb_hip.so")
CSRC = os.path.join(_HERE, "csrc")


class VdbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vdb error {code}: {msg}")
        self.code = code


def build(force=False, jobs=8):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def load():
    """Load the shared library (does not touch the GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VdbError(-100, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.vdb_last_error.restype = ctypes.c_char_p
        _lib.vdb_version.restype = ctypes.c_char_p
    return _lib


def check(rc):
    if rc != 0:
        raise VdbError(rc, load().vdb_last_error().decode())


_inited = None


def init(device=None):
    """Bind this process to one GPU (LOCAL_RANK by default).  Raises if no GPU is visible."""
    global _inited
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _inited != device:
        check(load().vdb_init(int(device)))
        _inited = device
    return load()
