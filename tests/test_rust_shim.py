"""integration/rust/src/hip_ffi.rs against include/adsb_hip.h, without a Rust toolchain.

The shim cannot be compiled in this image (no rustc / cargo), so nothing would notice if a prototype
in the header changed and the `unsafe extern "C"` block did not.  This test parses both files and
compares them mechanically: every function (name, arity, each argument's type, the return type), the
status constants, and the four structs (field order, types, offsets, sizes -- also against the
ctypes structs the Python host uses).  tests/abi_host.c pins the same layouts from the C side with
_Static_assert.
"""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "adsb_hip.h"
FFI = ROOT / "integration" / "rust" / "src" / "hip_ffi.rs"
BUILD_RS = ROOT / "integration" / "rust" / "build.rs"

# C base type -> Rust type
BASE = {
    "int": "c_int", "void": "c_void", "char": "c_char", "size_t": "usize", "double": "f64", "float": "f32",
    "int16_t": "i16", "uint16_t": "u16", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8",
    "adsb_ctx": "AdsbCtx", "adsb_msg": "AdsbMsg", "adsb_trial": "AdsbTrial", "adsb_stats": "AdsbStats",
    "adsb_multi": "AdsbMulti", "adsb_multi_stats": "AdsbMultiStats",
}
RUST_SIZE = {"u8": 1, "i16": 2, "u16": 2, "i32": 4, "u32": 4, "f32": 4, "u64": 8, "f64": 8}


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def c_type_to_rust(ctype: str) -> str:
    """`const adsb_ctx *` -> `*const AdsbCtx`, `int16_t **` -> `*mut *mut i16`, `size_t` -> `usize`."""
    toks = ctype.replace("*", " * ").split()
    const = bool(toks) and toks[0] == "const"
    if const:
        toks = toks[1:]
    base = toks[0]
    assert base in BASE, f"unknown C type {ctype!r}"
    out = BASE[base]
    # every `*` points at what stands to its left: the base type (const or not), or a pointer that is itself
    # const when a `const` follows its star (`const void *const *` = `*const *const c_void`)
    pointee_const, rest = const, toks[1:]
    while rest:
        assert rest[0] == "*", ctype
        out = ("*const " if pointee_const else "*mut ") + out
        rest = rest[1:]
        pointee_const = bool(rest) and rest[0] == "const"
        if pointee_const:
            rest = rest[1:]
    return out


def header_functions():
    text = strip_comments(HEADER.read_text())
    funcs = {}
    for m in re.finditer(r"^\s*((?:const\s+)?\w+\s*\**)\s*(adsb_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)(\w+)$", a)           # the last identifier is the parameter's name
                params.append(c_type_to_rust(mm.group(1).strip()))
        funcs[name] = (params, None if ret == "void" else c_type_to_rust(ret))
    return funcs


def rust_functions():
    text = strip_comments(FFI.read_text())
    m = re.search(r'unsafe\s+extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S)
    assert m, 'no `unsafe extern "C"` block in hip_ffi.rs'
    funcs = {}
    for f in re.finditer(r"pub\s+fn\s+(\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", m.group(1), flags=re.S):
        name, args, ret = f.group(1), " ".join(f.group(2).split()), f.group(3)
        params = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if a.strip()]
        funcs[name] = (params, " ".join(ret.split()) if ret else None)
    return funcs


def header_structs():
    text = strip_comments(HEADER.read_text())
    text = text.replace("ADSB_MODES_LONG_MSG_BYTES", "14")
    out = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(adsb_\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"^(\w+)\s+(\w+)(?:\[(\w+)\])?$", decl)
            assert mm, decl
            rt = BASE[mm.group(1)]
            fields.append((mm.group(2), f"[{rt}; {mm.group(3)}]" if mm.group(3) else rt))
        out[BASE[m.group(2)]] = fields
    return out


def rust_structs():
    text = strip_comments(FFI.read_text())
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub\s+struct\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(","):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"^(?:pub\s+)?(\w+)\s*:\s*(.+)$", decl)
            assert mm, decl
            fields.append((mm.group(1), mm.group(2).strip()))
        out[m.group(1)] = fields
    return out


def layout(fields):
    """(offsets, size) of a #[repr(C)] struct of the given Rust field types."""
    off, offsets, align_max = 0, [], 1
    for _, t in fields:
        arr = re.match(r"^\[(\w+); (\d+)\]$", t)
        elem, count = (arr.group(1), int(arr.group(2))) if arr else (t, 1)
        a = RUST_SIZE[elem]
        off = (off + a - 1) // a * a
        offsets.append(off)
        off += a * count
        align_max = max(align_max, a)
    return offsets, (off + align_max - 1) // align_max * align_max


def test_every_prototype_of_the_header_is_bound_identically():
    hdr, ffi = header_functions(), rust_functions()
    assert len(hdr) >= 30 and "adsb_demodulate2400" in hdr and "adsb_version" in hdr
    assert sorted(ffi) == sorted(hdr), (sorted(set(hdr) - set(ffi)), sorted(set(ffi) - set(hdr)))
    for name, (params, ret) in hdr.items():
        assert ffi[name][0] == params, f"{name}: header {params} vs hip_ffi.rs {ffi[name][0]}"
        assert ffi[name][1] == ret, f"{name}: returns {ret} in the header, {ffi[name][1]} in hip_ffi.rs"


def test_the_comparison_does_notice_a_changed_prototype(tmp_path, monkeypatch):
    """The check is not vacuous: a header whose adsb_collect gained a parameter, or whose adsb_to_mag
    takes a different pointer type, no longer matches hip_ffi.rs."""
    import sys
    mod = sys.modules[__name__]
    text = HEADER.read_text()
    for old, new in (("int adsb_collect(adsb_ctx *ctx, adsb_msg *out, size_t cap, size_t *n_out);",
                      "int adsb_collect(adsb_ctx *ctx, adsb_msg *out, size_t cap, size_t *n_out, int flags);"),
                     ("int adsb_to_mag(adsb_ctx *ctx, const int16_t *iq_re_im,", "int adsb_to_mag(adsb_ctx *ctx, const uint16_t *iq_re_im,"),
                     ("int adsb_pending(const adsb_ctx *ctx);", "size_t adsb_pending(const adsb_ctx *ctx);")):
        assert old in text
        fake = tmp_path / "adsb_hip.h"
        fake.write_text(text.replace(old, new))
        monkeypatch.setattr(mod, "HEADER", fake)
        with pytest.raises(AssertionError):
            test_every_prototype_of_the_header_is_bound_identically()
    monkeypatch.setattr(mod, "HEADER", ROOT / "include" / "adsb_hip.h")
    test_every_prototype_of_the_header_is_bound_identically()


def test_status_constants_match_the_header():
    hdr = dict(re.findall(r"^\s*(ADSB_(?:OK|ERR_\w+))\s*=\s*(-?\d+)", strip_comments(HEADER.read_text()), flags=re.M))
    ffi = dict(re.findall(r"pub const (ADSB_\w+): c_int = (-?\d+);", FFI.read_text()))
    assert hdr == ffi and len(hdr) == 9
    from dump1090_rs_amd import _lib
    assert {k: int(v) for k, v in hdr.items()} == {k: getattr(_lib, k) for k in hdr}


def test_struct_layouts_match_header_and_ctypes():
    from dump1090_rs_amd import _lib
    hdr, ffi = header_structs(), rust_structs()
    ctypes_of = {"AdsbMsg": _lib.AdsbMsg, "AdsbTrial": _lib.AdsbTrial, "AdsbStats": _lib.AdsbStats,
                 "AdsbMultiStats": _lib.AdsbMultiStats}
    want_size = {"AdsbMsg": 40, "AdsbTrial": 32, "AdsbStats": 72, "AdsbMultiStats": 96}
    for name, ct in ctypes_of.items():
        assert ffi[name] == hdr[name], f"{name}: field order / types differ between hip_ffi.rs and the header"
        offsets, size = layout(ffi[name])
        assert size == C.sizeof(ct) == want_size[name]
        assert [f[0] for f in ct._fields_] == [f[0] for f in ffi[name]]
        assert offsets == [getattr(ct, f[0]).offset for f in ct._fields_]
    assert ffi["AdsbCtx"] == [("_private", "[u8; 0]")]          # opaque
    assert ffi["AdsbMulti"] == [("_private", "[u8; 0]")]


def test_build_script_is_inert_without_the_hip_feature():
    """The reference's default `cargo build` / `cross test` must keep working with build.rs copied in:
    it may only ask for ADSB_HIP_DIR behind CARGO_FEATURE_HIP."""
    text = strip_comments(BUILD_RS.read_text())
    gate = text.index("CARGO_FEATURE_HIP")
    assert "return" in text[gate:text.index("ADSB_HIP_DIR\")", gate)]
    assert text.index("expect(") > gate and text.index("rustc-link-lib") > gate
