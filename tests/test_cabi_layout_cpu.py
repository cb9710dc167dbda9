"""CPU suite: the C-ABI structs are mirrored by hand in three places - include/*.h, the ctypes classes of
candle-video_amd/ltxhip/__init__.py and the #[repr(C)] definitions of rust/ltxhip-sys/src/lib.rs.  This test compiles
tests/cabi_layout.c as C99 against the headers (proving they are plain C), runs it, and checks sizeof / alignment /
every offsetof against the ctypes mirrors and against the LAYOUT_* constants the Rust file asserts at compile time.
It also checks that every `extern "C"` declaration of the Rust crate names a symbol the library exports."""
import ctypes
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def c_layout(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cabi") / "cabi_layout")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cabi_layout.c"), "-o", exe], check=True)
    return json.loads(subprocess.run([exe], check=True, capture_output=True, text=True).stdout)


def test_ctypes_mirrors_match_the_c_headers(c_layout):
    import ltxhip
    mirrors = {"ltx_weight": ltxhip._Weight, "ltx_dit_config": ltxhip.DitConfigC, "ltx_vae_config": ltxhip.VaeConfigC, "ltx_tiling": ltxhip.TilingC,
               "ltx_pipeline_params": ltxhip.PipelineParamsC, "ltx_t5_config": ltxhip.T5ConfigC}
    for name, cls in mirrors.items():
        c = c_layout[name]
        assert ctypes.sizeof(cls) == c["size"], (name, ctypes.sizeof(cls), c["size"])
        assert ctypes.alignment(cls) == c["align"], name
        py_fields = [f[0] for f in cls._fields_]
        assert py_fields == list(c["fields"].keys()), (name, py_fields, list(c["fields"].keys()))
        for f in py_fields:
            assert getattr(cls, f).offset == c["fields"][f], (name, f)


def test_rust_layout_constants_and_field_order_match_the_c_headers(c_layout):
    src = open(os.path.join(ROOT, "rust", "ltxhip-sys", "src", "lib.rs")).read()
    consts = dict(re.findall(r"pub const LAYOUT_(\w+): \(usize, usize\) = \((\d+), (\d+)\);", src) and
                  [(m[0], (int(m[1]), int(m[2]))) for m in re.findall(r"pub const LAYOUT_(\w+): \(usize, usize\) = \((\d+), (\d+)\);", src)])
    assert len(consts) == 6
    for key, (size, align) in consts.items():
        c = c_layout[key.lower()]
        assert (size, align) == (c["size"], c["align"]), (key, size, align, c)
        # the constant is tied to the Rust struct by a compile-time assertion
        assert re.search(r"size_of::<%s>\(\) == LAYOUT_%s\.0" % (key.lower(), key), src), key
    # field ORDER of every #[repr(C)] struct equals the header's (types are all 4- or 8-byte scalars / arrays of them)
    for name in ("ltx_weight", "ltx_dit_config", "ltx_vae_config", "ltx_tiling", "ltx_pipeline_params", "ltx_t5_config"):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, src, re.S).group(1)
        rust_fields = re.findall(r"pub (\w+):", body)
        assert rust_fields == list(c_layout[name]["fields"].keys()), (name, rust_fields)


def test_rust_extern_declarations_name_exported_symbols():
    src = open(os.path.join(ROOT, "rust", "ltxhip-sys", "src", "lib.rs")).read()
    fns = re.findall(r"pub fn (ltx_\w+)\(", src)
    assert len(fns) >= 30
    lib = ctypes.CDLL(os.path.join(ROOT, "candle-video_amd", "libltxhip.so"))
    for f in fns:
        assert hasattr(lib, f), f
    shim = open(os.path.join(ROOT, "rust", "hip_backend.rs")).read()
    used = set(re.findall(r"sys::(ltx_\w+)\(", shim))
    assert used and used <= set(fns), used - set(fns)          # the shim calls only what the sys crate declares
    for need in ("impl VideoTransformer3D for HipDit", "impl VaeLtxVideo for HipVae", "impl Drop for HipDit", "impl Drop for HipVae",
                 "pub fn from_files", "pub fn new("):
        assert need in shim, need
