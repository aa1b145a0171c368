"""The C-ABI library loads and exports every symbol include/lcr.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

from longcallr_amd import _abi, _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()  # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "lcr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lcr_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.lcr_version().startswith(b"liblcr")


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include "lcr.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(lcr_reads),sizeof(lcr_regions),sizeof(lcr_params),sizeof(lcr_candidate),sizeof(lcr_fragmat),'
                   'offsetof(lcr_candidate,loglik),offsetof(lcr_params,seed),sizeof(lcr_read_filter),offsetof(lcr_read_filter,divergence));return 0;}')
    exe = tmp_path / "sz"
    assert os.system("gcc -I%s %s -o %s" % (os.path.join(ROOT, "include"), src, exe)) == 0
    got = [int(x) for x in os.popen(str(exe)).read().split()]
    want = [C.sizeof(_abi.LcrReads), C.sizeof(_abi.LcrRegions), C.sizeof(_abi.LcrParams), _abi.CAND_DTYPE.itemsize,
            C.sizeof(_abi.LcrFragmat), _abi.CAND_DTYPE.fields["loglik"][1], _abi.LcrParams.seed.offset,
            C.sizeof(_abi.LcrReadFilter), _abi.LcrReadFilter.divergence.offset]
    assert got == want


def test_presets_match_main_rs(lib):
    for name, pid in _abi.PRESET_IDS.items():
        p = _abi.LcrParams()
        assert lib.lcr_params_preset(pid, C.byref(p)) == 0
        q = _abi.make_params(name)
        for f, _ in _abi.LcrParams._fields_:
            assert getattr(p, f) == getattr(q, f), (name, f)
    assert lib.lcr_params_preset(7, C.byref(_abi.LcrParams())) != 0


def test_fails_loudly_without_a_gpu(lib):
    """No CPU fallback: on a box without a HIP device the engine raises instead of computing."""
    import torch
    from longcallr_amd import api
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.LcrError, match="no CPU fallback"):
        api.Engine(0)


def test_integration_md_rust_block_matches_the_header(tmp_path):
    """The `#[repr(C)]` structs a maintainer would paste from INTEGRATION.md have the sizes and field offsets of
    include/lcr.h (checked by laying the Rust fields out with C rules and comparing with offsetof / sizeof from gcc),
    and every `pub fn` of the block is a symbol the header declares."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = md[md.index("```rust"):md.index("```", md.index("```rust") + 7)]
    block = re.sub(r"//[^\n]*", "", block)
    prim = {"i8": (1, 1), "u8": (1, 1), "i32": (4, 4), "u32": (4, 4), "f32": (4, 4), "i64": (8, 8), "u64": (8, 8), "f64": (8, 8)}

    def layout(fields):
        off, align, out = 0, 1, {}
        for name, ty in fields:
            ty = ty.strip()
            m = re.fullmatch(r"\[(\w+);\s*(\d+)\]", ty)
            if ty.startswith("*"):
                sz, al = 8, 8
            elif m:
                sz, al = prim[m.group(1)][0] * int(m.group(2)), prim[m.group(1)][1]
            else:
                sz, al = prim[ty]
            off = (off + al - 1) // al * al
            out[name] = off
            off += sz
            align = max(align, al)
        return out, (off + align - 1) // align * align
    structs = {}
    for m in re.finditer(r"pub struct (lcr_\w+)\s*\{(.*?)\}", block, flags=re.S):
        body = m.group(2)
        if "_p:" in body:
            continue       # opaque handles
        fields = [("ref" if a.strip() == "ref_" else a.strip(), b) for a, b in re.findall(r"pub (\w+):\s*([^,}]+)", body)]
        structs[m.group(1)] = layout(fields)
    want = ["lcr_reads", "lcr_regions", "lcr_params", "lcr_columns", "lcr_candidate", "lcr_candidate_list", "lcr_fragmat",
            "lcr_phase_result", "lcr_phase_collected", "lcr_region_list", "lcr_read_filter", "lcr_read_record"]
    assert sorted(structs) == sorted(want)
    lines = ['#include "lcr.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(){"]
    for name in want:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for f in structs[name][0]:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    lines.append("return 0;}")
    src = tmp_path / "off.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "off"
    assert os.system("gcc -I%s %s -o %s" % (os.path.join(ROOT, "include"), src, exe)) == 0
    got = dict(l.split() for l in os.popen(str(exe)).read().strip().split("\n"))
    for name in want:
        offs, size = structs[name]
        assert int(got[name]) == size, name
        for f, o in offs.items():
            assert int(got["%s.%s" % (name, f)]) == o, (name, f)
    fns = set(re.findall(r"pub fn (lcr_\w+)", block))
    assert fns <= set(_lib.SYMBOLS) and {"lcr_get_fragmat", "lcr_get_columns", "lcr_discover_regions", "lcr_get_ld_blocks"} <= fns
