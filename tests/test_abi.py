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
