#!/usr/bin/env python3
"""Check rust/eccoxide-gpu/src/ffi.rs against include/eccx.h, mechanically.

There is no Rust toolchain in the build image, so nothing compiles the Rust crate there; this
script is what keeps its `extern "C"` block honest.  It parses both files and fails on
  * a function declared on one side only,
  * a different number of parameters, a different parameter name, or a parameter / return type that
    is not the Rust spelling of the C type,
  * an enum constant of the header that is missing from ffi.rs or has another value,
  * an `ffi::name` used by the wrapper modules that ffi.rs does not declare,
  * a macro fragment of kind `path` or `ty` that the macro body then extends with `::` (rustc rejects
    `use $p::{..}` for `$p:path`: the construct that sat in weierstrass.rs through round 2),
  * an `eccoxide::...` path segment, an imported eccoxide item, or a method called on the crate's values that the
    reference does not define (against rust/eccoxide-gpu/reference_items.json, the public item NAMES extracted from
    the reference as text by tools/extract_rust_items.py),
  * unbalanced brackets in any source file of the crate (a cheap guard against truncated edits).

    python tools/check_rust_ffi.py          # exit code 0 = in sync
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "eccx.h")
CRATE = os.path.join(ROOT, "rust", "eccoxide-gpu")
FFI = os.path.join(CRATE, "src", "ffi.rs")

C_TO_RUST = {
    "int": "c_int",
    "size_t": "usize",
    "uint32_t": "u32",
    "const uint8_t*": "*const u8",
    "uint8_t*": "*mut u8",
    "const void*": "*const c_void",
    "void*": "*mut c_void",
    "eccx_ctx*": "*mut eccx_ctx",
    "const eccx_ctx*": "*const eccx_ctx",
    "eccx_ctx**": "*mut *mut eccx_ctx",
    "const char*": "*const c_char",
    "void": None,
}


def strip_c_comments(txt):
    return re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)


def norm_c_type(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\s*\*", "*", t)
    return t


def parse_header(path=None):
    txt = strip_c_comments(open(path or HEADER).read())
    funcs = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(eccx_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = norm_c_type(m.group(1)), m.group(2), m.group(3)
        params = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            if a == "void":
                continue
            mm = re.match(r"(.*?)([A-Za-z_]\w*)$", a)
            params.append((mm.group(2), norm_c_type(mm.group(1))))
        funcs[name] = (ret, params)
    consts = {}
    for body in re.findall(r"enum\s*\w*\s*\{(.*?)\}", txt, flags=re.S):
        nxt = 0
        for item in [x.strip() for x in body.split(",") if x.strip()]:
            if "=" in item:
                k, v = [s.strip() for s in item.split("=", 1)]
                nxt = eval(re.sub(r"(\d+)u\b", r"\1", v), {"__builtins__": {}}, {})
            else:
                k = item
            consts[k] = nxt
            nxt += 1
    return funcs, consts


def strip_rust_comments(txt):
    return re.sub(r"//[^\n]*", "", txt)


def parse_ffi(path=None):
    txt = strip_rust_comments(open(path or FFI).read())
    ext = re.search(r'extern\s+"C"\s*\{(.*)\}', txt, flags=re.S).group(1)
    funcs = {}
    for m in re.finditer(r"pub\s+fn\s+(eccx_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ext, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "").strip() or None
        params = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            k, t = [s.strip() for s in a.split(":", 1)]
            params.append((k, re.sub(r"\s+", " ", t)))
        funcs[name] = (ret, params)
    consts = {}
    for m in re.finditer(r"pub\s+const\s+(ECCX_\w+)\s*:\s*\w+\s*=\s*([^;]+);", txt):
        consts[m.group(1)] = eval(m.group(2), {"__builtins__": {}}, {})
    return funcs, consts


def balanced(path):
    txt = strip_rust_comments(open(path).read())
    txt = re.sub(r'"(\\.|[^"\\])*"', '""', txt)      # string literals
    txt = re.sub(r"'(\\.|[^'\\])'", "''", txt)       # char literals (lifetimes have no closing quote)
    stack = []
    pairs = {")": "(", "]": "[", "}": "{"}
    for ch in txt:
        if ch in "([{":
            stack.append(ch)
        elif ch in pairs:
            if not stack or stack.pop() != pairs[ch]:
                return False
    return not stack


# methods of std / core types the wrappers call; anything else called with `.name(` must be an eccoxide item
STD_METHODS = {
    "iter", "enumerate", "map", "collect", "zip", "copied", "flatten", "unwrap", "try_into", "extend_from_slice", "len", "push",
    "as_ptr", "as_mut_ptr", "chunks_exact", "to_string_lossy", "into_owned", "is_empty", "clone", "extend", "take", "map_or",
    "raw", "check", "opts", "opts_var", "id", "sizes", "point", "is_rejected", "device", "to_vec", "into_iter", "get", "any", "all",
    "filter", "unwrap_or", "ok_or", "cloned", "as_slice", "first", "last", "rev", "sum", "fmt", "to_string", "join", "display",
    "exists", "is_some", "is_none", "as_ref", "as_mut", "as_deref",
}


def lint_wrappers(files):
    """Lints that need no compiler: fragment misuse in macro_rules, and names against the reference's item list."""
    errors = []
    items_path = os.path.join(CRATE, "reference_items.json")
    ref = json.load(open(items_path))["files"] if os.path.exists(items_path) else None
    names = set()
    mods = {"eccoxide", "curve", "protocol"}
    if ref:
        for f, kinds in ref.items():
            for kind, vals in kinds.items():
                names |= set(vals)
                if kind == "mod":
                    mods |= set(vals)
    for path in files:
        rel = os.path.relpath(path, ROOT)
        txt = strip_rust_comments(open(path).read())
        # (1) `$x:path` / `$x:ty` fragments followed by `::` in the expansion
        for mac in re.finditer(r"macro_rules!\s*\w+\s*\{(.*?)\n\}", txt, flags=re.S):
            body = mac.group(1)
            for frag, kind in re.findall(r"\$(\w+):(path|ty)\b", body):
                if re.search(r"\$" + frag + r"\s*::", body):
                    errors.append(f"{rel}: macro fragment ${frag}:{kind} is extended with `::` (not accepted by rustc; match `$($seg:ident)::+` instead)")
        if ref is None:
            continue
        # (2) eccoxide paths: every module segment and every imported item must exist in the reference
        for m in re.finditer(r"\beccoxide((?:::\w+)+)(?:::\{([^}]*)\})?", txt):
            segs = [x for x in m.group(1).split("::") if x]
            leaf_items = [x.strip() for x in (m.group(2) or "").split(",") if x.strip()]
            if not leaf_items:
                leaf_items, segs = [segs[-1]], segs[:-1]
            for sgm in segs:
                if sgm not in mods and sgm not in names:
                    errors.append(f"{rel}: eccoxide::...::{sgm}: no such module in the reference's item list")
            for it in leaf_items:
                if it not in names and it not in mods:
                    errors.append(f"{rel}: eccoxide item {it} is not in the reference's item list")
        # (3) methods and associated functions
        for meth in sorted(set(re.findall(r"\.([a-z_][a-z0-9_]*)\(", txt))):
            if meth not in STD_METHODS and meth not in names:
                errors.append(f"{rel}: method .{meth}() is neither a known std method nor an item of the reference")
        for typ, fn in sorted(set(re.findall(r"\b([A-Z]\w+)::([a-z_][a-z0-9_]*)\(", txt))):
            if typ in ("Vec", "Self", "GpuError", "GpuContext", "Unit", "CStr", "String", "Some", "Ok", "Err", "Curve", "Secrecy", "Bases"):
                continue
            if fn not in names:
                errors.append(f"{rel}: {typ}::{fn}() is not an item of the reference")
        for typ, var in sorted(set(re.findall(r"\b(Sign)::([A-Z]\w+)", txt))):
            if f"{typ}::{var}" not in names:
                errors.append(f"{rel}: enum variant {typ}::{var} is not in the reference's item list")
    return errors


def check():
    errors = []
    c_funcs, c_consts = parse_header()
    r_funcs, r_consts = parse_ffi()
    for name in sorted(set(c_funcs) | set(r_funcs)):
        if name not in r_funcs:
            errors.append(f"{name}: declared in eccx.h, missing from ffi.rs")
            continue
        if name not in c_funcs:
            errors.append(f"{name}: declared in ffi.rs, not in eccx.h")
            continue
        (c_ret, c_par), (r_ret, r_par) = c_funcs[name], r_funcs[name]
        if c_ret not in C_TO_RUST:
            errors.append(f"{name}: C return type {c_ret!r} has no Rust mapping in this checker")
        elif C_TO_RUST[c_ret] != r_ret:
            errors.append(f"{name}: returns {c_ret!r} in C, {r_ret!r} in Rust")
        if len(c_par) != len(r_par):
            errors.append(f"{name}: {len(c_par)} parameters in C, {len(r_par)} in Rust")
            continue
        for i, ((cn, ct), (rn, rt)) in enumerate(zip(c_par, r_par)):
            if ct not in C_TO_RUST:
                errors.append(f"{name}: parameter {i} has C type {ct!r} with no Rust mapping in this checker")
            elif C_TO_RUST[ct] != rt:
                errors.append(f"{name}: parameter {i} ({cn}) is {ct!r} in C but {rt!r} in Rust")
            if cn != rn:
                errors.append(f"{name}: parameter {i} is named {cn!r} in C and {rn!r} in Rust")
    for k, v in sorted(c_consts.items()):
        if k not in r_consts:
            errors.append(f"{k}: enum constant of eccx.h missing from ffi.rs")
        elif r_consts[k] != v:
            errors.append(f"{k}: {v} in eccx.h, {r_consts[k]} in ffi.rs")
    for k in sorted(set(r_consts) - set(c_consts)):
        errors.append(f"{k}: constant of ffi.rs that eccx.h does not define")
    declared = set(r_funcs) | set(r_consts) | {"eccx_ctx"}
    src = os.path.join(CRATE, "src")
    files = [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith(".rs")] + [os.path.join(CRATE, "build.rs")]
    for path in files:
        if not balanced(path):
            errors.append(f"{os.path.relpath(path, ROOT)}: unbalanced brackets")
        if path.endswith("ffi.rs") or path.endswith("build.rs"):
            continue
        for used in sorted(set(re.findall(r"ffi::((?:eccx|ECCX)_\w+)", strip_rust_comments(open(path).read())))):
            if used not in declared:
                errors.append(f"{os.path.relpath(path, ROOT)}: uses ffi::{used}, which ffi.rs does not declare")
    errors += lint_wrappers([f for f in files if not f.endswith("ffi.rs") and not f.endswith("build.rs")])
    return errors, len(c_funcs), len(c_consts)


def main():
    errors, nf, nc = check()
    if errors:
        print("\n".join(errors))
        sys.exit(1)
    print(f"ffi.rs and eccx.h agree: {nf} functions, {nc} constants")


if __name__ == "__main__":
    main()
