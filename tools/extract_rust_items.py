#!/usr/bin/env python3
"""Extract, AS TEXT, the public item names of the reference files the Rust wrappers lean on.

    python tools/extract_rust_items.py > rust/eccoxide-gpu/reference_items.json      (needs /root/reference)

The build image has no Rust toolchain, so nothing resolves the `eccoxide::...` paths and the methods that
rust/eccoxide-gpu/src/*.rs name.  This script lists what the reference DEFINES -- `pub fn`, `pub struct`, `pub enum`,
`pub trait`, `pub mod`, `pub const`, `pub type`, and the same inside `macro_rules!` bodies (most of the surface is
macro-generated: src/curve/fiat/field_macros.rs, curve_macros.rs) -- per file, names only; tools/check_rust_ffi.py then
checks every eccoxide item and method the wrappers use against the committed list.  Names are data about the
reference's API surface, not source text.
"""
import json
import os
import re
import sys

REF = "/root/reference"
FILES = ["src/lib.rs", "src/curve/mod.rs", "src/curve/group.rs", "src/curve/field.rs", "src/curve/affine.rs", "src/curve/projective.rs",
         "src/curve/fiat/field_macros.rs", "src/curve/fiat/curve_macros.rs", "src/curve/sec2/mod.rs", "src/curve/sec2/p256r1.rs",
         "src/curve/sec2/p384r1.rs", "src/curve/sec2/p521r1.rs", "src/curve/curve25519.rs", "src/curve/bls12_381/mod.rs",
         "src/curve/bls12_381/g1.rs", "src/curve/bls12_381/fp.rs", "src/curve/bls12_381/scalar.rs", "src/curve/bls12_381/serialize.rs",
         "src/protocol/x25519.rs", "src/protocol/ed25519.rs"]
PAT = re.compile(r"\bpub(?:\([a-z ]+\))?\s+(?:const\s+)?(fn|struct|enum|trait|mod|const|type|use)\s+([A-Za-z_][A-Za-z0-9_]*)")


def main():
    out = {}
    for rel in FILES:
        path = os.path.join(REF, rel)
        if not os.path.exists(path):
            continue
        txt = re.sub(r"//[^\n]*", "", open(path).read())
        items = {}
        for kind, name in PAT.findall(txt):
            if kind != "use":
                items.setdefault(kind, set()).add(name)
        # enum variants the wrappers match on (Sign::Positive / Negative)
        for m in re.finditer(r"pub\s+enum\s+(\w+)\s*\{([^}]*)\}", txt):
            items.setdefault("variant", set()).update(f"{m.group(1)}::{v}" for v in re.findall(r"\b([A-Z]\w*)\b\s*(?:,|$|\()", m.group(2)))
        # types named by a macro invocation: fiat_field_montgomery_impl!(#[doc = ".."] FieldElement, ...)
        for m in re.finditer(r"\bfiat_\w+!\(\s*(?:#\[[^\]]*\]\s*)*([A-Z]\w*)\s*,", txt):
            items.setdefault("macro_type", set()).add(m.group(1))
        # re-exports: pub use a::b::{C, D}
        for m in re.finditer(r"pub\s+use\s+([^;]+);", txt):
            items.setdefault("reexport", set()).update(re.findall(r"\b([A-Za-z_]\w*)\b", m.group(1).split("::")[-1]))
        out[rel] = {k: sorted(v) for k, v in sorted(items.items())}
    json.dump({"source": "public item NAMES of vincenthz/eccoxide (text scan, tools/extract_rust_items.py)", "files": out}, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
