"""ctypes binding of the C-ABI library (include/mhaq_fq.h -> mhaq_amd/csrc/libmhaq_fq.so).

There is NO fallback: if the library is missing or a call fails, this raises.  The
product path never routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# MHAQ_FQ_LIB: an A/B build of the same library (tools/variants.sh) for kernel tuning runs; never a fallback
LIB_PATH = os.environ.get("MHAQ_FQ_LIB") or os.path.join(_HERE, "csrc", "libmhaq_fq.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mhaq_fq.h")

_p = C.c_void_p
_i64 = C.c_int64
_u64 = C.c_uint64
_sz = C.c_size_t
_int = C.c_int

# name -> (restype, argtypes); must list every function declared in include/mhaq_fq.h
# (tests/test_capi_symbols.py cross-checks this table against the header).
SIGNATURES = {
    "mhaq_fq_abi_version": (_int, []),
    "mhaq_fq_error_string": (C.c_char_p, [_int]),
    "mhaq_fq_fill_r": (_int, [_p, _i64, _u64, _u64, _p]),
    "mhaq_fq_pt_fwd_workspace_bytes": (_sz, [_i64]),
    "mhaq_fq_pt_fwd": (_int, [_p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "mhaq_fq_pt_bwd_workspace_bytes": (_sz, [_i64]),
    "mhaq_fq_pt_bwd": (_int, [_p, _p, _p, _i64, _p, _p, _p, _p, _int, _p, _i64, _p, _u64, _u64, _p, _int, _p, _p, _sz, _p]),
    "mhaq_fq_pt_bwd_partials": (_int, [_p, _p, _p, _i64, _p, _p, _p, _p, _int, _p, _i64, _p, _u64, _u64, _p, _int, _p, _sz,
                                       _p, _p]),
    "mhaq_fq_pt_bwd_finalize": (_int, [_p, C.c_int32, _p, _p]),
    "mhaq_fq_act_fwd": (_int, [_p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "mhaq_fq_act_bwd_workspace_bytes": (_sz, [_i64]),
    "mhaq_fq_act_bwd": (_int, [_p, _p, _p, _i64, _p, _int, _p, _u64, _u64, _p, _p, _p, _sz, _p]),
    "mhaq_fq_act_bwd_partials": (_int, [_p, _p, _p, _i64, _p, _int, _p, _u64, _u64, _p, _p, _sz, _p, _p]),
    "mhaq_fq_act_bwd_finalize_multi": (_int, [_p, _int, _p, _p]),
    "mhaq_fq_minmax_workspace_bytes": (_sz, [_i64]),
    "mhaq_fq_minmax": (_int, [_p, _i64, _p, _p, _sz, _p]),
    "mhaq_fq_row_minmax": (_int, [_p, _i64, _i64, _p, _p, _p]),
    "mhaq_fq_pt_tie_scatter": (_int, [_p, _p, _i64, _p, _p, _p]),
    "mhaq_fq_pt_aewgs_colstats_workspace_bytes": (C.c_size_t, [_i64, _i64]),
    "mhaq_fq_pt_aewgs_colstats": (_int, [_p, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, C.c_size_t, _p]),
    "mhaq_fq_pc_fwd": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _p]),
    "mhaq_fq_pc_bwd": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, _u64, _u64, _p, _p]),
    "mhaq_fq_pc_aewgs_stats": (_int, [_p, _p, _p, _p, _i64, _i64, _p, _p]),
    "mhaq_fq_wlayer_fwd": (_int, [_p, _p, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    "mhaq_fq_wlayer_bwd": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _int, _p, _p, _p, _u64, _u64, _p, _p]),
    "mhaq_fq_wlayer_fwd_multi": (_int, [_p, _int, _i64, _i64, _p, _p, _p]),
    "mhaq_fq_wlayer_bwd_multi": (_int, [_p, _int, _i64, _i64, _p, _p, _p, _int, _p, _u64, _u64, _p, _p]),
    "mhaq_fq_wlayer_bwd_group": (_int, [_p, _int, _i64, _i64, _p, _i64, _p, _p, _int, _p, _u64, _u64, _p, _p]),
    "mhaq_fq_wlayer_aewgs_stats_group": (_int, [_p, _int, _i64, _p, _i64, _p, _p]),
    "mhaq_fq_pc_quantize": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _p, _p]),
    "mhaq_fq_wlayer_pt_max_elements": (_i64, []),
    "mhaq_fq_wlayer_pt_fwd": (_int, [_p, _p, _p, _i64, _p, _p]),
    "mhaq_fq_wlayer_pt_bwd": (_int, [_p, _p, _p, _p, _p, _p, _i64, _int, _p, _u64, _u64, _p, _p]),
    "mhaq_fq_wlayer_ptl_workspace_bytes": (_sz, [_i64]),
    "mhaq_fq_wlayer_ptl_fwd": (_int, [_p, _p, _p, _i64, _p, _p, _sz, _p]),
    "mhaq_fq_wlayer_ptl_bwd": (_int, [_p, _p, _p, _p, _p, _p, _i64, _int, _p, _i64, _p, _u64, _u64, _p, _p, _sz, _p]),
    "mhaq_fq_vec_fwd": (_int, [_p, _p, _p, _p, _p, _i64, _p]),
    "mhaq_fq_vec_aewgs_stats": (_int, [_p, _p, _p, _p, _i64, _p, _p]),
    "mhaq_fq_vec_bwd": (_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _int, _p, _p, _u64, _u64, _p, _p]),
    "mhaq_fq_potential_loss_fwd": (_int, [_p, _p, _p, _i64, _p, _p, _i64, C.c_float, C.c_float, C.c_float, _int,
                                          _p, _int, _p, _p]),
    "mhaq_fq_potential_loss_bwd": (_int, [_p, _p, _p, _p, _i64, _p, _p, _i64, C.c_float, C.c_float, C.c_float,
                                          _p, _p, _p, _p, _p, _p]),
    "mhaq_fq_noise_fwd": (_int, [_p, _p, _i64, _p]),
    "mhaq_fq_noise_bwd_workspace_bytes": (_sz, [_i64, _i64]),
    "mhaq_fq_noise_bwd": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _i64, _p, _u64, _u64, _p, _p, _sz, _p]),
}

_lib = None


class MhaqFqError(RuntimeError):
    pass


def header_functions():
    """Names of the functions declared in include/mhaq_fq.h."""
    with open(HEADER_PATH) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(mhaq_fq_\w+)\s*\(", src)))


def _try_build() -> None:
    """A fresh checkout has no .so (built artefacts are git-ignored): compile it once with hipcc.  This is
    still the HIP path -- if the toolchain is absent the caller gets the error below, never a fallback.
    Under torch.distributed.run every rank gets here at once: the build runs under an exclusive file lock
    (the ranks that lose the race wait, then find the library), and the Makefile renames a finished file into
    place, so no rank ever maps a partially written library."""
    import fcntl
    import subprocess
    import sys
    csrc = os.path.dirname(LIB_PATH)
    with open(os.path.join(csrc, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(LIB_PATH):
                return
            print(f"[mhaq_amd] {LIB_PATH} missing: running `make -C {csrc} libmhaq_fq.so`", file=sys.stderr, flush=True)
            try:
                subprocess.run(["make", "-C", csrc, "libmhaq_fq.so"], check=True, stdout=subprocess.DEVNULL)
            except (OSError, subprocess.CalledProcessError) as e:
                print(f"[mhaq_amd] build failed: {e}", file=sys.stderr, flush=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            _try_build()
        if not os.path.exists(LIB_PATH):
            raise MhaqFqError(
                f"{LIB_PATH} is missing: build it with `make -C mhaq_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`.  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if L.mhaq_fq_abi_version() != 4:
            raise MhaqFqError("libmhaq_fq.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mhaq_fq_error_string(rc).decode()
        raise MhaqFqError(f"{what} failed: {msg} (code {rc})")
