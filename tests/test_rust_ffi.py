"""rust/eccoxide-gpu/src/ffi.rs is the header, symbol for symbol (tools/check_rust_ffi.py).  CPU only:
no Rust toolchain is needed -- or available -- to keep the two in step."""
import importlib.util
import os
import re

from tests.oracle_lib import ROOT

spec = importlib.util.spec_from_file_location("check_rust_ffi", os.path.join(ROOT, "tools", "check_rust_ffi.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def test_ffi_rs_matches_the_header():
    errors, nfuncs, nconsts = chk.check()
    assert errors == []
    from eccoxide_amd import _lib

    assert nfuncs == len(_lib.SYMBOLS)          # and the ctypes table binds the same set


def test_the_checker_sees_drift(tmp_path):
    """Every kind of mismatch the checker exists for is actually reported."""
    good = open(chk.FFI).read()
    c_funcs, c_consts = chk.parse_header()

    def errors_for(text):
        p = tmp_path / "ffi.rs"
        p.write_text(text)
        r_funcs, r_consts = chk.parse_ffi(str(p))
        saved = chk.FFI
        chk.FFI = str(p)
        try:
            return chk.check()[0]
        finally:
            chk.FFI = saved

    assert errors_for(good) == []
    # wrong pointer constness
    bad = good.replace("pub fn eccx_comb_table(ctx: *mut eccx_ctx, curve: c_int, out: *mut u8)",
                       "pub fn eccx_comb_table(ctx: *mut eccx_ctx, curve: c_int, out: *const u8)")
    assert bad != good and any("eccx_comb_table" in e and "parameter 2" in e for e in errors_for(bad))
    # dropped parameter
    bad = good.replace("pub fn eccx_prepare(ctx: *mut eccx_ctx, curve: c_int, what: u32)",
                       "pub fn eccx_prepare(ctx: *mut eccx_ctx, curve: c_int)")
    assert any("eccx_prepare" in e and "parameters" in e for e in errors_for(bad))
    # wrong return type
    bad = good.replace("pub fn eccx_device_bytes(ctx: *const eccx_ctx) -> usize;", "pub fn eccx_device_bytes(ctx: *const eccx_ctx) -> c_int;")
    assert any("eccx_device_bytes" in e and "returns" in e for e in errors_for(bad))
    # missing function
    bad = re.sub(r"pub fn eccx_x25519_dev\(.*?;", "", good, flags=re.S)
    assert any("eccx_x25519_dev" in e and "missing" in e for e in errors_for(bad))
    # constant with another value
    bad = good.replace("pub const ECCX_CT_SCAN: u32 = 1 << 8;", "pub const ECCX_CT_SCAN: u32 = 1 << 9;")
    assert any("ECCX_CT_SCAN" in e for e in errors_for(bad))
    # swapped argument names (same types): u1 / u2 of the verify shape
    bad = good.replace("u1: *const u8, u2: *const u8,\n                                 q: *const u8", "u2: *const u8, u1: *const u8,\n                                 q: *const u8")
    assert bad != good and any("eccx_double_scalarmul" in e and "named" in e for e in errors_for(bad))
    assert len(c_funcs) >= 27 and c_consts["ECCX_ASSUME_SUBGROUP"] == 1 << 9


def test_wrapper_modules_cover_the_reference_surface():
    """The batch functions the crate promises exist, per curve."""
    src = os.path.join(ROOT, "rust", "eccoxide-gpu", "src")
    w = open(os.path.join(src, "weierstrass.rs")).read()
    for fn in ("mul_batch", "mul_base_batch", "verify_points", "add_batch", "decompress_batch", "compress_batch"):
        assert f"pub fn {fn}(" in w
    for curve in ("p256r1", "p384r1", "p521r1"):
        assert f"gpu_weierstrass_curve!({curve}," in w
    b = open(os.path.join(src, "bls12_381_g1.rs")).read()
    for fn in ("mul_batch", "mul_base_batch", "from_compressed_batch", "to_compressed_batch", "is_in_subgroup_batch"):
        assert f"pub fn {fn}(" in b
    e = open(os.path.join(src, "ed25519.rs")).read()
    for fn in ("scale_batch", "mul_base_batch", "verify_points", "decode_points", "encode_points"):
        assert f"pub fn {fn}(" in e
    assert "pub fn x25519_batch(" in open(os.path.join(src, "x25519.rs")).read()


def _lint(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return chk.lint_wrappers([str(p)])


def test_the_lints_catch_what_a_compiler_would(tmp_path):
    """No rustc here: the lints that stand in for it must fire on the constructs they exist for."""
    src = os.path.join(ROOT, "rust", "eccoxide-gpu", "src")
    w = open(os.path.join(src, "weierstrass.rs")).read()
    assert _lint(tmp_path, "weierstrass.rs", w) == []
    # round 2's construct: a `path` fragment extended with `::{...}` (rejected by rustc)
    bad = w.replace("$($seg:ident)::+,", "$path:path,").replace("use $($seg)::+::{FieldElement, PointAffine, Scalar};",
                                                                    "use $path::{FieldElement, PointAffine, Scalar};")
    assert bad != w and any("$path:path" in e for e in _lint(tmp_path, "weierstrass.rs", bad))
    # a method eccoxide does not have
    bad = w.replace("p.to_coordinate()", "p.to_coordinates()")
    assert any("to_coordinates" in e for e in _lint(tmp_path, "weierstrass.rs", bad))
    # an associated function / item / module / enum variant that does not exist
    assert any("from_coords" in e for e in _lint(tmp_path, "weierstrass.rs", w.replace("PointAffine::from_coordinate(", "PointAffine::from_coords(")))
    assert any("Sign::Odd" in e for e in _lint(tmp_path, "weierstrass.rs", w.replace("Sign::Negative", "Sign::Odd")))
    b = open(os.path.join(src, "bls12_381_g1.rs")).read()
    assert _lint(tmp_path, "bls12_381_g1.rs", b) == []
    assert any("bls12381" in e for e in _lint(tmp_path, "bls12_381_g1.rs", b.replace("eccoxide::curve::bls12_381::g1::PointAffine", "eccoxide::curve::bls12381::g1::PointAffine")))
    assert any("FpElement" in e for e in _lint(tmp_path, "bls12_381_g1.rs", b.replace("use eccoxide::curve::bls12_381::{Fp, Scalar};", "use eccoxide::curve::bls12_381::{FpElement, Scalar};")))


def test_secrecy_has_no_default_and_results_keep_the_three_flags():
    """The reference's `*` and mul_base are constant-time: a drop-in caller must choose, and a rejected
    unit must not look like the point at infinity."""
    lib = open(os.path.join(ROOT, "rust", "eccoxide-gpu", "src", "lib.rs")).read()
    sec = lib[lib.index("pub enum Secrecy"):]
    head = lib[:lib.index("pub enum Secrecy")].rstrip().splitlines()[-1]
    assert "Default" not in head and "#[default]" not in sec[:sec.index("}")]
    assert "Default::default()" not in lib
    assert "pub enum Unit<P>" in lib and all(v in lib for v in ("Point(P)", "Infinity", "Rejected"))
    for f in ("weierstrass.rs", "bls12_381_g1.rs", "ed25519.rs"):
        txt = open(os.path.join(ROOT, "rust", "eccoxide-gpu", "src", f)).read()
        assert "Vec<Option<Point" not in txt.replace("&[Option<PointAffine>]", "") and "Unit::from_flag" in txt


def test_wrapper_buffer_sizes_match_the_library():
    """The byte layouts the wrappers allocate (FB, SB, encodings) are the sizes the library reports for each curve."""
    from eccoxide_amd import _lib

    lib = _lib.load()
    w = open(os.path.join(ROOT, "rust", "eccoxide-gpu", "src", "weierstrass.rs")).read()
    ids = {"p256r1": 0, "p384r1": 1, "p521r1": 2}
    for curve, fb, sb in re.findall(r"gpu_weierstrass_curve!\((\w+), [\w:]+, [\w:]+, (\d+), (\d+)\);", w):
        assert lib.eccx_field_bytes(ids[curve]) == int(fb) and lib.eccx_scalar_bytes(ids[curve]) == int(sb)
        assert lib.eccx_compressed_bytes(ids[curve]) == int(fb) + 1      # compress_batch returns [u8; FB + 1]
    b = open(os.path.join(ROOT, "rust", "eccoxide-gpu", "src", "bls12_381_g1.rs")).read()
    assert "const FB: usize = 48;" in b and "const SB: usize = 32;" in b
    assert lib.eccx_field_bytes(3) == 48 and lib.eccx_scalar_bytes(3) == 32 and lib.eccx_compressed_bytes(3) == 48
    assert lib.eccx_field_bytes(4) == 32 and lib.eccx_scalar_bytes(4) == 32 and lib.eccx_compressed_bytes(4) == 32
