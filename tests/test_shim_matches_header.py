"""There is no rustc in this build environment, so the Rust side of the boundary is pinned textually: the raw
bindings (rust/biogpu-sys/src/lib.rs) must declare exactly the functions of include/biogpu.h — same names, same
arity, same integer widths / pointer constness — the same struct layouts and constants, and the library must export
every one of them.  The shim crate (rust/bio-gpu-shim) may only call functions that exist."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_sys as gen  # noqa: E402

RS = os.path.join(ROOT, "rust", "biogpu-sys", "src", "lib.rs")


def rust_functions():
    src = open(RS).read()
    block = src[src.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block, flags=re.S):
        params = [tuple(s.strip() for s in p.split(":", 1)) for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = (params, (m.group(3) or "c_void").strip())
    return out


def rust_structs():
    src = open(RS).read()
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\}", src, flags=re.S):
        fields = [tuple(s.strip() for s in f.replace("pub ", "").split(":", 1)) for f in m.group(2).split(",\n") if ":" in f]
        out[m.group(1)] = fields
    return out


def test_generated_file_is_current():
    assert open(RS).read() == gen.generate(), "rust/biogpu-sys/src/lib.rs is stale: run python tools/gen_rust_sys.py"


def test_symbols_arity_and_widths():
    _, funcs, _, _ = gen.parse_header()
    rs = rust_functions()
    assert set(rs) == {f[0] for f in funcs}
    for name, ret, params in funcs:
        r_params, r_ret = rs[name]
        assert len(r_params) == len(params), name
        assert r_ret == gen.rust_type(ret), name
        for (c_name, c_type), (r_name, r_type) in zip(params, r_params):
            assert r_name.replace("r#", "") == c_name, (name, c_name)
            assert r_type == gen.rust_type(c_type), (name, c_name)
            # integer widths, spelled out: the generator must not be the only witness
            for c_int, width in (("uint64_t", "u64"), ("uint32_t", "u32"), ("int32_t", "i32"), ("uint8_t", "u8"), ("int64_t", "i64")):
                if c_type.replace("const ", "").replace("*", "").strip() == c_int:
                    assert r_type.split()[-1] == width, (name, c_name)


def test_struct_layouts():
    structs, _, _, _ = gen.parse_header()
    rs = rust_structs()
    for nm, fields in structs.items():
        assert nm in rs, nm
        assert [f[0] for f in rs[nm]] == [f[0] for f in fields], nm
    # sizes the device kernels rely on
    from rust_bio_amd import _lib
    width = {"i8": 1, "u8": 1, "i32": 4, "u32": 4, "f32": 4, "i64": 8, "u64": 8}

    def size(ty):
        if ty in rs:  # nested struct (bg_seed_hit_t.aln)
            return sum(size(t) for _, t in rs[ty])
        arr = re.match(r"\[(\w+); (\d+)\]", ty)
        if arr:
            return width[arr.group(1)] * int(arr.group(2))
        return 8 if ty.startswith("*") else width[ty]

    assert sum(size(t) for _, t in rs["bg_alignment_t"]) == _lib.ALN_DTYPE.itemsize == 64
    assert sum(size(t) for _, t in rs["bg_fastq_record_t"]) == _lib.FQREC_DTYPE.itemsize == 56
    assert sum(size(t) for _, t in rs["bg_seed_hit_t"]) == _lib.SEED_HIT_DTYPE.itemsize == 96


def test_constants_match():
    _, _, enums, defines = gen.parse_header()
    src = open(RS).read()
    for nm, v in enums:
        assert re.search(rf"pub const {nm}: c_int = {v};", src), nm
    assert "pub const BG_MIN_SCORE: i32 = -858993459;" in src


def test_library_exports_every_bound_symbol_and_shim_calls_exist():
    from rust_bio_amd import _lib
    rs = rust_functions()
    assert set(rs) == set(_lib.SYMBOLS)
    L = _lib.lib()
    for name in rs:
        assert hasattr(L, name), name
    shim = os.path.join(ROOT, "rust", "bio-gpu-shim", "src")
    for f in os.listdir(shim):
        for call in re.findall(r"sys::(bg_\w+)\(", open(os.path.join(shim, f)).read()):
            assert call in rs, (f, call)
    # AlignmentMode is mapped explicitly, never `mode as i32`
    assert "as i32" not in re.sub(r"is_some\(\) as i32", "", open(os.path.join(shim, "pairwise.rs")).read().replace("o as i32", ""))
