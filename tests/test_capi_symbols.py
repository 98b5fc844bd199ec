"""CPU: the C-ABI library loads and exports every symbol include/mhaq_fq.h declares; argument
validation works without a GPU (no compute calls are made here)."""
import ctypes
import os
import re
import subprocess

import pytest

from mhaq_amd import _lib


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.dirname(_lib.LIB_PATH)], check=True)
    return _lib.lib()


def test_every_declared_function_is_exported_and_bound(L):
    declared = _lib.header_functions()
    assert len(declared) >= 17
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and include/mhaq_fq.h disagree"
    for name in declared:
        assert hasattr(L, name), f"{name} missing from libmhaq_fq.so"


def test_argument_counts_match_header():
    src = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER_PATH).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else len([p for p in params.split(",") if p.strip()])
        assert n == len(args), f"{name}: header has {n} parameters, ctypes table {len(args)}"


def test_abi_version_and_error_strings(L):
    assert L.mhaq_fq_abi_version() == 4
    assert L.mhaq_fq_error_string(0) == b"ok"
    assert b"invalid" in L.mhaq_fq_error_string(-1)
    assert b"workspace" in L.mhaq_fq_error_string(-2)


def test_argument_errors_are_reported_not_thrown(L):
    # null pointers / bad method / short workspace are rejected before any launch
    assert L.mhaq_fq_pt_fwd(None, None, 16, None, None, None, None, None, None, None, None, 0, None) == -1
    assert L.mhaq_fq_pt_bwd(None, None, None, -1, None, None, None, None, 0, None, 0, None, 0, 0, None, 0, None, None, 0, None) == -1
    assert L.mhaq_fq_pc_bwd(None, None, None, None, None, None, 4, 0, 0, None, None, None, 0, 0, None, None) == -1
    assert L.mhaq_fq_minmax(None, 0, None, None, 0, None) == -1
    fake = ctypes.c_void_p(0x1000)
    assert L.mhaq_fq_pt_bwd(fake, fake, fake, 8, fake, fake, fake, fake, 7, None, 0, None, 0, 0, None, 0, fake, fake, 1 << 20, None) == -1
    assert L.mhaq_fq_pt_bwd(fake, fake, fake, 8, fake, fake, fake, fake, 0, None, 0, None, 0, 0, None, 0, fake, fake, 8, None) == -2
    assert L.mhaq_fq_pt_fwd(ctypes.c_void_p(0x1001), fake, 8, fake, fake, fake, fake, None, None, None, None, 0, None) == -3
    assert L.mhaq_fq_pt_bwd_workspace_bytes(1 << 20) >= 64 * 5 * 4
    # the grouped weight backward: null table, no layers, a window stride shorter than the group, unknown estimator
    assert L.mhaq_fq_wlayer_bwd_group(None, 1, 4, 4, fake, 4, fake, fake, 0, None, 0, 0, None, None) == -1
    assert L.mhaq_fq_wlayer_bwd_group(fake, 0, 4, 4, fake, 4, fake, fake, 0, None, 0, 0, None, None) == -1
    assert L.mhaq_fq_wlayer_bwd_group(fake, 1, 4, 4, fake, 3, fake, fake, 0, None, 0, 0, None, None) == -1
    assert L.mhaq_fq_wlayer_bwd_group(fake, 1, 4, 4, fake, 4, fake, fake, 9, None, 0, 0, None, None) == -1
    assert L.mhaq_fq_wlayer_aewgs_stats_group(fake, 1, 4, fake, 4, None, None) == -1
    assert L.mhaq_fq_wlayer_aewgs_stats_group(fake, 1, 0, fake, 4, fake, None) == -1
    # quantize with given per-row parameters (ABI v4): null parameters / tensors are argument errors, an empty tensor is fine
    assert L.mhaq_fq_pc_quantize(fake, fake, None, None, fake, 4, 9, None, None) == -1
    assert L.mhaq_fq_pc_quantize(None, fake, None, fake, fake, 4, 9, None, None) == -1
    assert L.mhaq_fq_pc_quantize(None, None, None, fake, fake, 0, 9, None, None) == 0
    # the stand-alone noise backward runs one workgroup per group on grid.x: more groups than HIP's grid bound are unsupported
    assert L.mhaq_fq_noise_bwd(fake, fake, fake, fake, (1 << 24), 1, 0, None, 0, None, 0, 0, None, fake, 1 << 40, None) == -4
    # the streaming per-tensor weight layer: empty tensor, null pointers, short workspace, misaligned weight,
    # unknown estimator / AEWGS without statistics (rejected by the streaming backward before any launch)
    nb = L.mhaq_fq_wlayer_ptl_workspace_bytes(1 << 16)
    assert nb >= L.mhaq_fq_minmax_workspace_bytes(1 << 16) and nb >= L.mhaq_fq_pt_bwd_workspace_bytes(1 << 16) + 28
    assert L.mhaq_fq_wlayer_ptl_fwd(fake, fake, fake, 0, fake, fake, nb, None) == -1
    assert L.mhaq_fq_wlayer_ptl_fwd(None, fake, fake, 1 << 16, fake, fake, nb, None) == -1
    assert L.mhaq_fq_wlayer_ptl_fwd(fake, fake, fake, 1 << 16, fake, fake, 16, None) == -2
    assert L.mhaq_fq_wlayer_ptl_fwd(ctypes.c_void_p(0x1002), fake, fake, 1 << 16, fake, fake, nb, None) == -3
    assert L.mhaq_fq_wlayer_ptl_bwd(fake, fake, fake, None, fake, None, 1 << 16, 0, None, 0, None, 0, 0, None, fake, nb, None) == -1
    assert L.mhaq_fq_wlayer_ptl_bwd(fake, fake, fake, fake, fake, None, 1 << 16, 0, None, 0, None, 0, 0, None, fake, 16, None) == -2
    assert L.mhaq_fq_wlayer_ptl_bwd(fake, fake, fake, fake, fake, None, 1 << 16, 5, None, 0, None, 0, 0, None, fake, nb, None) == -1
    assert L.mhaq_fq_wlayer_ptl_bwd(fake, fake, fake, fake, fake, None, 1 << 16, 2, None, 0, None, 0, 0, None, fake, nb, None) == -1


def test_header_is_plain_c_and_cxx():
    """include/mhaq_fq.h is the drop-in boundary: it must compile as C99 (the FFI of any host language binds C) and as
    C++11, with no dependency beyond <stddef.h> / <stdint.h>."""
    import shutil
    import subprocess
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mhaq_fq.h")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr], check=True)
    src = open(hdr).read()
    includes = [ln.strip() for ln in src.splitlines() if ln.strip().startswith("#include")]
    assert sorted(includes) == ["#include <stddef.h>", "#include <stdint.h>"]


def test_the_ctypes_stub_documented_in_integration_md_declares_the_librarys_signatures(L):
    """INTEGRATION.md section 3 shows a maintainer a ctypes stub: its argtypes must be the ones this package binds (the
    full sample -- stub + autograd.Function -- is executed on the GPU by tests/test_gpu_integration_doc.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## 3. Bind the C ABI directly"):text.index("## 4.")]
    code = re.findall(r"```python\n(.*?)```", sec, flags=re.S)[0]
    stub = code[:code.index("class FakeQuantAct")].replace('C.CDLL("libmhaq_fq.so")', f'C.CDLL("{_lib.LIB_PATH}")')
    ns = {}
    exec(compile(stub, "INTEGRATION.md section 3 (stub)", "exec"), ns)
    bound = 0
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(ns["_L"], name)
        if fn.argtypes is None:
            continue                                  # the sample binds only what its Function calls
        bound += 1
        assert [a._type_ for a in fn.argtypes] == [a._type_ for a in args], name
        if fn.restype is not ctypes.c_int:            # ctypes' default: what every int-returning entry point needs
            assert fn.restype._type_ == res._type_, name
    assert bound >= 3
