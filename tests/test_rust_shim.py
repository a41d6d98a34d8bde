"""CPU: the Rust shim under integration/rust/ against include/radiorust_amd.h.

There is no Rust toolchain in this image, so the shim cannot be compiled; what can be checked mechanically is
  * every entry point the header declares is declared in ffi.rs with the same name, arity, argument types
    and return type (C type -> Rust type by the table below), and nothing else is declared;
  * the #[repr(C)] structs have the header's fields in the header's order, the enums the header's values;
  * every `ffi::rr_*` call in the block modules names a declared function and passes as many arguments as it takes;
  * ffi.rs is what scripts/gen_rust_ffi.py generates from the current header.
The parsers here are independent of the generator's."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "radiorust_amd.h")
GPU_DIR = os.path.join(ROOT, "integration", "rust", "src", "blocks", "gpu")
FFI = os.path.join(GPU_DIR, "ffi.rs")

C2RUST = {"int": "c_int", "double": "f64", "float": "f32", "size_t": "usize", "uint64_t": "u64", "int64_t": "i64",
          "uint32_t": "u32", "int32_t": "i32", "char": "c_char", "void": "c_void"}


def c_text():
    return re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)


def c_type_to_rust(t):
    toks = t.replace("*", " * ").split()
    ptr = toks.count("*")
    const = "const" in toks
    base = [x for x in toks if x not in ("*", "const", "struct")]
    assert len(base) == 1, t
    r = C2RUST.get(base[0], base[0])
    for level in range(ptr):
        r = ("*const " if const and level == 0 else "*mut ") + r
    return r


def header_functions():
    out = {}
    for stmt in c_text().split(";"):
        stmt = " ".join(stmt.split())
        m = re.search(r"(\brr_[a-z0-9_]+)\s*\((.*)\)$", stmt)
        if not m or "typedef" in stmt:
            continue
        name, args = m.group(1), m.group(2).strip()
        ret = stmt[: m.start(1)].strip()
        ret = ret.split("{")[-1].split("}")[-1].strip()  # what precedes the first declaration (extern "C" {, #endif ..)
        ret = " ".join(w for w in ret.split() if not w.startswith("#") and w not in ("RADIORUST_AMD_H",))
        ret = re.sub(r"^.*\b(?:endif|define \w+)\b", "", ret).strip()
        params = []
        if args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)(\w+)$", a)
                params.append(c_type_to_rust(mm.group(1)))
        out[name] = (params, None if ret == "void" else c_type_to_rust(ret))
    return out


def split_top_level(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts]


def rust_functions():
    src = open(FFI).read()
    src = re.sub(r"//.*", "", src)
    out = {}
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", src, flags=re.S):
        name, args, ret = m.group(1), m.group(2), m.group(3)
        params = []
        for a in split_top_level(" ".join(args.split())):
            if a:
                params.append(a.split(":", 1)[1].strip())
        out[name] = (params, ret.strip() if ret else None)
    return out


def test_ffi_rs_declares_exactly_the_header():
    h, r = header_functions(), rust_functions()
    assert len(h) > 100  # the parser found the header's functions
    assert sorted(h) == sorted(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    for name in h:
        assert h[name] == r[name], (name, h[name], r[name])


def test_ffi_rs_structs_and_enums_match_the_header():
    text, src = c_text(), open(FFI).read()
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            first = re.match(r"^(.*?)(\w+(?:\s*,\s*\w+)*)$", decl)
            for n in first.group(2).replace(" ", "").split(","):
                fields.append((n, c_type_to_rust(first.group(1))))
        rs = re.search(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct %s \{(.*?)\}" % m.group(2), src, flags=re.S)
        assert rs, m.group(2)
        got = [(a, b.strip()) for a, b in re.findall(r"pub (\w+): ([^,]+),", rs.group(1))]
        assert got == fields, (m.group(2), got, fields)
    for m in re.finditer(r"enum\s+\w+\s*\{(.*?)\}", text, flags=re.S):
        nxt = 0
        for it in m.group(1).split(","):
            it = it.strip()
            if not it:
                continue
            if "=" in it:
                k, v = [s.strip() for s in it.split("=")]
                nxt = int(v, 0)
            else:
                k = it
            assert re.search(r"pub const %s: c_int = %d;" % (k, nxt), src), (k, nxt)
            nxt += 1
    # every handle type of the header is an opaque #[repr(C)] struct
    for h in re.findall(r"typedef\s+struct\s+(rr_\w+)\s+\1\s*;", text):
        assert re.search(r"#\[repr\(C\)\]\s*pub struct %s \{" % h, src), h


def test_block_modules_call_declared_functions_with_the_right_arity():
    r = rust_functions()
    used = set()
    for f in sorted(os.listdir(GPU_DIR)):
        if not f.endswith(".rs") or f == "ffi.rs":
            continue
        src = re.sub(r"//.*", "", open(os.path.join(GPU_DIR, f)).read())
        for m in re.finditer(r"ffi::(rr_\w+)\b", src):
            name = m.group(1)
            if name not in r:
                # a type (ffi::rr_c64 { .. }, *mut ffi::rr_block): must be a struct of ffi.rs
                assert re.search(r"pub struct %s\b" % name, open(FFI).read()), (f, name)
                continue
            rest = src[m.end():].lstrip()
            if not rest.startswith("("):
                used.add(name)  # passed as a function pointer (the handle's destroy)
                continue
            depth, i = 0, 0
            for i, ch in enumerate(rest):
                depth += ch in "([{"
                depth -= ch in ")]}"
                if depth == 0:
                    break
            args = split_top_level(" ".join(rest[1:i].split()))
            assert len([a for a in args if a]) == len(r[name][0]), (f, name, args, r[name][0])
            used.add(name)
    # the blocks of the path are all there
    for need in ("rr_freqshifter_enqueue", "rr_filter_design", "rr_filter_enqueue", "rr_filter_reset", "rr_downsampler_peek",
                 "rr_downsampler_enqueue", "rr_fourier_set_sampled_window", "rr_fourier_enqueue", "rr_chain_enqueue",
                 "rr_chain_interrupt", "rr_host_register", "rr_wait"):
        assert need in used, need


def test_ffi_rs_is_in_sync_with_its_generator():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_rust_ffi.py"), "--check"])
    assert p.returncode == 0, "regenerate: python scripts/gen_rust_ffi.py"


def test_shim_mirrors_the_reference_api_names():
    """Same type, constructor and method names as the CPU blocks (transform.rs:282-296,376-390; filters.rs:128-152,
    279-297; resampling.rs:38-50; analysis.rs:39-59)."""
    want = {
        "freq_shifter.rs": ["pub struct FreqShifter<Flt>", "pub fn new()", "pub fn with_shift(", "pub fn with_precision(",
                            "pub fn with_precision_and_shift(", "pub fn precision(", "pub fn shift(", "pub fn set_shift(",
                            "pub fn update_shift<"],
        "filter.rs": ["pub struct Filter<Flt>", "pub fn new<F>(", "pub fn new_rectangular<F>(", "pub fn with_window<F, W>(",
                      "pub fn update<F>(", "pub fn update_with_window<F, W>("],
        "downsampler.rs": ["pub struct Downsampler<Flt>", "pub fn new(output_chunk_len: usize, output_rate: f64, bandwidth: f64)",
                           "pub fn with_quality("],
        "fourier.rs": ["pub struct Fourier<Flt>", "pub fn new()", "pub fn new_center_dc()", "pub fn with_window<W>(",
                       "pub fn with_window_center_dc<W>("],
    }
    for f, items in want.items():
        src = open(os.path.join(GPU_DIR, f)).read()
        for it in items:
            assert it in src, (f, it)
        assert "impl_block_trait! { <Flt> Consumer<Signal<Complex<Flt>>>" in src
        assert "impl_block_trait! { <Flt> Producer<Signal<Complex<Flt>>>" in src
