"""The C-ABI library loads and exports every symbol include/ani_abi.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "ani_abi.h")
LIB = os.path.join(ROOT, "fastani_amd", "csrc", "libfastani_amd.so")


def declared():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ani_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return LIB


def test_header_symbols_exported(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    exported = set(re.findall(r" T (ani_[a-z_0-9]+)", out))
    names = declared()
    assert len(names) >= 20
    missing = [n for n in names if n not in exported]
    assert not missing, missing


def test_library_loads_and_reports_no_device(lib):
    """dlopen resolves every HIP / rocPRIM dependency; without a GPU ani_init must fail loudly (no CPU fallback)."""
    import fastani_amd.api as api
    L = ctypes.CDLL(lib)
    api._bind(L)
    h = ctypes.c_void_p()
    rc = L.ani_init(0, ctypes.byref(h))
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        assert rc != 0 and b"no CPU path" in L.ani_last_error()
    else:
        assert rc == 0
        L.ani_shutdown(h)


def test_pod_layouts():
    import fastani_amd.api as api
    assert api.MINIMIZER_DT.itemsize == 12       # skch::MinimizerInfo   base_types.hpp:22
    assert api.MAPPING_DT.itemsize == 44         # skch::MappingResult   base_types.hpp:89
    assert api.CGI_DT.itemsize == 20             # cgi::CGI_Results      cgid_types.hpp:68
    assert ctypes.sizeof(api.Params) == 16
    assert ctypes.sizeof(api.SeqBatch) == 48
    # the python mirror of ani_counters_t must have one field per header field
    src = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    body = re.search(r"typedef struct \{([^}]*)\} ani_counters_t;", src, flags=re.S).group(1)
    n_fields = sum(len(decl.split(",")) for decl in re.findall(r"(?:uint64_t|double)\s+([^;]+);", body))
    assert n_fields == len(api.Counters._fields_)


def test_package_loader_refuses_without_library(tmp_path, monkeypatch):
    import fastani_amd._lib as L
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "missing.so"))
    monkeypatch.setattr(L, "_lib", None)
    with pytest.raises(ImportError):
        L.load()
