"""Loader of the compiled autograd binding (mhaq_amd/csrc/torch_binding.cpp -> mhaq_amd/csrc/_mhaq_torch.so).

The extension holds the torch::autograd::Function nodes of the fused layer ops; it calls the SAME C ABI as the ctypes
binding (include/mhaq_fq.h), resolved from the library path `_lib` uses.  Like the library it has no fallback: if it
is missing it is built once (hipcc, under the library's build lock), and if that fails the caller gets the error.
"""
from __future__ import annotations

import importlib.util
import os
import sys

from . import _lib

EXT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "_mhaq_torch.so")
_ext = None


def _try_build() -> None:
    import fcntl
    import subprocess
    csrc = os.path.dirname(EXT_PATH)
    with open(os.path.join(csrc, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(EXT_PATH):
                return
            print(f"[mhaq_amd] {EXT_PATH} missing: running `make -C {csrc} _mhaq_torch.so`", file=sys.stderr, flush=True)
            try:
                subprocess.run(["make", "-C", csrc, "_mhaq_torch.so"], check=True, stdout=subprocess.DEVNULL)
            except (OSError, subprocess.CalledProcessError) as e:
                print(f"[mhaq_amd] build failed: {e}", file=sys.stderr, flush=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def ext():
    """The bound extension module."""
    global _ext
    if _ext is None:
        import torch  # noqa: F401  (libtorch must be loaded before the extension resolves its symbols)
        _lib.lib()    # builds / validates the C-ABI library first
        if not os.path.exists(EXT_PATH):
            _try_build()
        if not os.path.exists(EXT_PATH):
            raise _lib.MhaqFqError(
                f"{EXT_PATH} is missing: build it with `make -C mhaq_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`.  There is no Python fallback.")
        spec = importlib.util.spec_from_file_location("_mhaq_torch", EXT_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.bind(_lib.LIB_PATH, _lib.MhaqFqError)
        _ext = mod
    return _ext
