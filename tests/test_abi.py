"""The C-ABI shared library loads and exports every symbol include/eccx.h declares; the
GPU-free entry points behave; the product fails loudly without a GPU.  CPU only."""
import ctypes
import os
import re

import pytest

from tests.oracle_lib import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "eccx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(eccx_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    from eccoxide_amd import _lib

    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/eccx.h but not exported by libeccx.so"
    assert set(syms) == set(_lib.SYMBOLS), "ctypes binding table and header disagree"


def test_sizes_without_gpu():
    import eccoxide_amd as E

    assert [E.field_bytes(c) for c in range(5)] == [32, 48, 66, 48, 32]
    assert [E.scalar_bytes(c) for c in range(5)] == [32, 48, 66, 32, 32]
    lib = E._lib.load()
    assert lib.eccx_field_bytes(99) == -1 and lib.eccx_scalar_bytes(-1) == -1
    assert lib.eccx_strerror(-1) == b"unknown curve id"
    assert lib.eccx_strerror(0) == b"ok"


def test_null_and_bad_arguments():
    from eccoxide_amd import _lib

    lib = _lib.load()
    assert lib.eccx_init(0, None) == -2
    # null context is rejected before anything touches a device
    assert lib.eccx_scalarmul_var(None, 0, 1, b"\0" * 32, b"\0" * 64, ctypes.create_string_buffer(64),
                                  ctypes.create_string_buffer(1), None, 0) == -2
    assert lib.eccx_scalarmul_base_dev(None, 0, 1, None, None, None, None, 0, None) == -2
    lib.eccx_shutdown(None)  # no-op


def test_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    import eccoxide_amd as E

    with pytest.raises(E.EccxError):
        E.Engine(0)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from eccoxide_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libeccx.so"))
    with pytest.raises(_lib.EccxLibraryMissing):
        _lib.load()


def test_product_does_not_reference_the_oracle():
    """No file of the product package imports, loads or links anything under oracle/."""
    pkg = os.path.join(ROOT, "eccoxide_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".cpp", ".hip", ".inc", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "ecc_ref" not in txt and "oracle/" not in txt.replace(
                    "never does", ""), f
