"""CPU-side checks of the product library: it loads, exports every symbol include/vdb.h declares,
and fails loudly (no CPU fallback) when no GPU is present.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from halo2_vectordb_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "vdb.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vdb_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol(lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_no_silent_cpu_fallback(lib):
    if lib.vdb_device_count() > 0:
        pytest.skip("GPU present")
    assert lib.vdb_init(0) == -6  # VDB_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.vdb_last_error()
    buf = (ctypes.c_uint64 * 4)()
    assert lib.vdb_fr_mul(buf, buf, buf, ctypes.c_size_t(1)) == -1  # VDB_ERR_NOT_INIT
    from halo2_vectordb_amd import api
    with pytest.raises(api.VdbError):
        api.fr_mul([[0, 0, 0, 0]], [[0, 0, 0, 0]])


def test_multi_device_lifecycle_without_a_gpu(lib):
    """b0: vdb_init_devices / vdb_set_device fail loudly too; shutdown is always safe"""
    if lib.vdb_device_count() > 0:
        pytest.skip("GPU present")
    assert lib.vdb_init_devices(1) == -6 and b"no CPU fallback" in lib.vdb_last_error()
    assert lib.vdb_set_device(0) == -3 and lib.vdb_current_device() == -1 and lib.vdb_devices_bound() == 0
    lib.vdb_shutdown()
    lib.vdb_shutdown()
    assert lib.vdb_init(0) == -6


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "halo2_vectordb_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.replace("oracle under /oracle is test infrastructure", ""), (dp, f)


def test_header_is_plain_c(tmp_path):
    """include/vdb.h is what a C / Rust-bindgen / cgo caller includes: it must compile as C99 on its own, warnings as errors"""
    import subprocess
    src = tmp_path / "h.c"
    src.write_text('#include "vdb.h"\nint main(void) { return vdb_device_count() < 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "h.o")])
