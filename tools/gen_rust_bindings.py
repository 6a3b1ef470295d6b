#!/usr/bin/env python3
"""Generates the Rust `extern "C"` binding of include/mpecdsa_hip.h — the raw `-sys` layer a maintainer of the reference
(ZenGo-X/multi-party-ecdsa, a Rust crate) adds to reach libmpecdsa_hip.so — FROM THE HEADER, so that the binding text can
never drift from the C-ABI: every exported function, every struct with every field in order, every constant.

    python tools/gen_rust_bindings.py            # rewrites tools/rust_shim/mpecdsa-hip-sys/src/lib.rs and the block between
                                                 # the BEGIN/END GENERATED markers of INTEGRATION.md
    python tools/gen_rust_bindings.py --check    # exit 1 if either is stale (tests/test_abi_cpu.py runs this)

The header is run through `gcc -E` (comments and include guards gone), split into top-level declarations and translated
type by type.  No Rust toolchain exists in this image: the output is text here, compiled by whoever integrates it."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mpecdsa_hip.h")
OUT_RS = os.path.join(ROOT, "tools", "rust_shim", "mpecdsa-hip-sys", "src", "lib.rs")
INTEGRATION = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED: tools/gen_rust_bindings.py -->", "<!-- END GENERATED -->"

SCALARS = {"int": "c_int", "uint32_t": "u32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "uint8_t": "u8",
           "size_t": "usize", "void": "c_void", "char": "c_char", "float": "f32", "unsigned": "c_uint"}
RUST_KEYWORDS = {"in", "type", "ref", "self", "fn", "mod", "move", "match", "loop", "use", "box", "as"}


def preprocess():
    src = subprocess.check_output(["gcc", "-E", "-P", "-x", "c", "-D__cplusplus_off", HEADER], text=True)
    # drop everything the system headers contributed: keep from the first mpe_ symbol's typedef on
    start = src.index("typedef struct mpe_ctx mpe_ctx;")
    return src[start:]


def constants():
    out = []
    for m in re.finditer(r"^#define\s+(MPE_\w+)\s+\(?(-?\d+)\)?", open(HEADER).read(), re.M):
        out.append((m.group(1), int(m.group(2))))
    return out


def split_decls(src):
    decls, depth, cur = [], 0, []
    for ch in src:
        cur.append(ch)
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        elif ch == ";" and depth == 0:
            d = " ".join("".join(cur).split())
            if d != ";":
                decls.append(d[:-1].strip())
            cur = []
    return decls


def rust_type(ctype):
    """'const uint32_t *' -> '*const u32' ; 'mpe_ctx * *' -> '*mut *mut mpe_ctx'"""
    t = ctype.replace("*", " * ").split()
    const = False
    base, ptrs = None, []
    for tok in t:
        if tok == "const":
            const = True
        elif tok == "struct":
            continue
        elif tok == "*":
            ptrs.append("*const " if const else "*mut ")
            const = False
        else:
            base = tok
    r = SCALARS.get(base, base)
    for p in ptrs:                      # innermost pointer first
        r = p + r
    return r


def ident(name):
    return "r#" + name if name in RUST_KEYWORDS else name


def parse_fields(body):
    """'uint32_t *z, *e; int kind; uint8_t ord[4];' -> [(name, rust type)]"""
    fields = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        first, *rest = [x.strip() for x in stmt.split(",")]
        m = re.match(r"(.*?)(\**)\s*(\w+)\s*(\[\d+\])?$", first)
        base = m.group(1).strip()
        for decl in [first[len(m.group(1)):]] + rest:
            dm = re.match(r"(\**)\s*(\w+)\s*(?:\[(\d+)\])?$", decl.strip())
            ty = rust_type(base + " " + dm.group(1))
            if dm.group(3):
                ty = f"[{ty}; {dm.group(3)}]"
            fields.append((dm.group(2), ty))
    return fields


def parse(src):
    opaque, structs, funcs = [], [], []
    for d in split_decls(src):
        m = re.match(r"typedef struct (\w+) (\w+)$", d)
        if m:
            opaque.append(m.group(2))
            continue
        m = re.match(r"typedef struct (?:\w+ )?\{(.*)\} (\w+)$", d)
        if m:
            structs.append((m.group(2), parse_fields(m.group(1))))
            continue
        m = re.match(r"(.*?)(\w+) ?\((.*)\)$", d)
        if not m:
            raise SystemExit(f"gen_rust_bindings: cannot parse declaration: {d[:120]}")
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                am = re.match(r"(.*?)(\w+)$", a.strip())
                params.append((am.group(2), rust_type(am.group(1))))
        funcs.append((name, params, None if ret == "void" else rust_type(ret)))
    return opaque, structs, funcs


def wrap(prefix, items, suffix, width=128, indent="        "):
    lines, cur = [], prefix
    for i, it in enumerate(items):
        piece = it + (", " if i + 1 < len(items) else "")
        if len(cur) + len(piece) > width and cur.strip():
            lines.append(cur.rstrip())
            cur = indent + piece
        else:
            cur += piece
    lines.append(cur + suffix)
    return "\n".join(lines)


def generate():
    opaque, structs, funcs = parse(preprocess())
    o = ["// GENERATED by tools/gen_rust_bindings.py from include/mpecdsa_hip.h — do not edit; re-run the script.",
         "// Raw bindings of libmpecdsa_hip.so (the MI355X batched crypto core).  Every pointer named d_* is a DEVICE pointer,",
         "// `stream` is a hipStream_t; the comments of the header (call sites of the reference each entry replaces) apply verbatim.",
         "#![allow(non_camel_case_types, non_snake_case)]",
         "use std::os::raw::{c_char, c_int, c_void};", ""]
    for name, val in constants():
        o.append(f"pub const {name}: c_int = {val};")
    o.append("")
    for name in opaque:
        o.append(f"#[repr(C)] pub struct {name} {{ _private: [u8; 0] }}")
    o.append("")
    for name, fields in structs:
        o.append("#[repr(C)] #[derive(Clone, Copy)]")
        o.append(wrap(f"pub struct {name} {{ ", [f"pub {ident(f)}: {t}" for f, t in fields], " }", indent="    "))
    o += ["", '#[link(name = "mpecdsa_hip")]', 'extern "C" {']
    for name, params, ret in funcs:
        tail = ")" + (f" -> {ret}" if ret else "") + ";"
        o.append(wrap(f"    pub fn {name}(", [f"{ident(p)}: {t}" for p, t in params], tail))
    o.append("}")
    return "\n".join(o) + "\n", opaque, structs, funcs


def main():
    check = "--check" in sys.argv
    text, opaque, structs, funcs = generate()
    block = f"{BEGIN}\n```rust\n{text}```\n{END}"
    doc = open(INTEGRATION).read()
    if BEGIN not in doc or END not in doc:
        raise SystemExit("INTEGRATION.md has no GENERATED markers")
    new_doc = doc[:doc.index(BEGIN)] + block + doc[doc.index(END) + len(END):]
    stale = []
    if not os.path.exists(OUT_RS) or open(OUT_RS).read() != text:
        stale.append(OUT_RS)
    if new_doc != doc:
        stale.append(INTEGRATION)
    if check:
        if stale:
            print("stale (run tools/gen_rust_bindings.py):", *stale)
            return 1
        return 0
    os.makedirs(os.path.dirname(OUT_RS), exist_ok=True)
    open(OUT_RS, "w").write(text)
    open(INTEGRATION, "w").write(new_doc)
    print(f"{len(funcs)} functions, {len(structs)} structs, {len(opaque)} opaque types -> {os.path.relpath(OUT_RS, ROOT)}, INTEGRATION.md")
    return 0


if __name__ == "__main__":
    sys.exit(main())
