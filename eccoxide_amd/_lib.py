"""ctypes binding of libeccx.so (the C ABI declared in include/eccx.h).

The library is the product: there is no Python or CPU fallback.  If it has not been
built (python -c 'import __graft_entry__ as g; g.build()' or make -C eccoxide_amd/csrc)
loading fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ECCX_LIB_PATH") or os.path.join(_HERE, "libeccx.so")  # override: A/B builds

# every symbol include/eccx.h declares, with its ctypes signature
_u8p = c_void_p  # buffers are passed as raw addresses (bytes, bytearray, numpy, torch data_ptr)
SYMBOLS = {
    "eccx_field_bytes": (c_int, [c_int]),
    "eccx_scalar_bytes": (c_int, [c_int]),
    "eccx_init": (c_int, [c_int, POINTER(c_void_p)]),
    "eccx_shutdown": (None, [c_void_p]),
    "eccx_last_error": (c_char_p, [c_void_p]),
    "eccx_strerror": (c_char_p, [c_int]),
    "eccx_prepare": (c_int, [c_void_p, c_int, c_uint32]),
    "eccx_reserve": (c_int, [c_void_p, c_int, c_size_t, c_uint32]),
    "eccx_device_bytes": (c_size_t, [c_void_p]),
    "eccx_scalarmul_var": (c_int, [c_void_p, c_int, c_size_t, _u8p, _u8p, _u8p, _u8p, _u8p, c_uint32]),
    "eccx_scalarmul_base": (c_int, [c_void_p, c_int, c_size_t, _u8p, _u8p, _u8p, _u8p, c_uint32]),
    "eccx_scalarmul_var_dev": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    "eccx_scalarmul_base_dev": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    "eccx_point_add": (c_int, [c_void_p, c_int, c_size_t, _u8p, _u8p, _u8p, _u8p, _u8p, _u8p, c_uint32]),
    "eccx_point_add_dev": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    "eccx_double_scalarmul": (c_int, [c_void_p, c_int, c_size_t, _u8p, _u8p, _u8p, _u8p, _u8p, c_uint32]),
    "eccx_x25519": (c_int, [c_void_p, c_size_t, _u8p, _u8p, _u8p, _u8p, c_uint32]),
    "eccx_compressed_bytes": (c_int, [c_int]),
    "eccx_point_decompress": (c_int, [c_void_p, c_int, c_size_t, _u8p, _u8p, _u8p, c_uint32]),
    "eccx_point_compress": (c_int, [c_void_p, c_int, c_size_t, _u8p, _u8p, _u8p, c_uint32]),
    "eccx_point_decompress_dev": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    "eccx_point_compress_dev": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    "eccx_double_scalarmul_dev": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    "eccx_x25519_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    "eccx_comb_table": (c_int, [c_void_p, c_int, _u8p]),
    "eccx_scalarmul_var_sharded": (c_int, [POINTER(c_void_p), c_int, c_int, c_size_t, _u8p, _u8p, _u8p, _u8p, c_uint32]),
    "eccx_scalarmul_base_sharded": (c_int, [POINTER(c_void_p), c_int, c_int, c_size_t, _u8p, _u8p, _u8p, c_uint32]),
}

_lib = None


class EccxLibraryMissing(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libeccx.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EccxLibraryMissing(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback."
        )
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  Whichever HIP runtime is
    # mapped first serves the whole process (same soname), and torch cannot find a GPU
    # when the system runtime got there first: import torch before dlopen-ing the engine
    # so both share torch's runtime.  (C/C++ hosts that never load torch use /opt/rocm's.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
