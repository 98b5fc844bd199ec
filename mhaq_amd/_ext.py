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


STAMP_PATH = EXT_PATH + ".stamp"


def expected_stamp():
    """"<torch version> <hash of torch_binding.cpp + mhaq_fq.h>": what the Makefile writes next to the extension it builds.
    None when the sources are not there (a binary-only deployment has nothing to compare against)."""
    import hashlib
    import importlib.metadata as md
    src = os.path.join(os.path.dirname(EXT_PATH), "torch_binding.cpp")
    if not (os.path.exists(src) and os.path.exists(_lib.HEADER_PATH)):
        return None
    h = hashlib.sha256(open(src, "rb").read() + open(_lib.HEADER_PATH, "rb").read()).hexdigest()[:16]
    return f"{md.version('torch')} {h}"


def _stamp_state() -> str:
    """"fresh": the stamp next to the extension names this torch and these sources (or there are no sources to compare
    against); "stale": it names another torch or other sources; "unknown": there is no stamp -- an extension built by an
    earlier Makefile, or shipped prebuilt next to its sources."""
    want = expected_stamp()
    if want is None:
        return "fresh"
    try:
        return "fresh" if open(STAMP_PATH).read().strip() == want else "stale"
    except OSError:
        return "unknown"


def _stale() -> bool:
    """The extension on disk is not known to match this torch and these sources: it is rebuilt before it is loaded."""
    return _stamp_state() != "fresh"


def _try_build(force=False) -> None:
    import fcntl
    import subprocess
    csrc = os.path.dirname(EXT_PATH)
    with open(os.path.join(csrc, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(EXT_PATH) and not (force and _stale()):
                return
            why = {"stale": "stale (built for another torch or from other sources)",
                   "unknown": "without a build stamp"}.get(_stamp_state(), "stale") if os.path.exists(EXT_PATH) else "missing"
            print(f"[mhaq_amd] {EXT_PATH} {why}: running `make -C {csrc} _mhaq_torch.so`", file=sys.stderr, flush=True)
            try:
                subprocess.run(["make"] + (["-B"] if force else []) + ["-C", csrc, "_mhaq_torch.so"], check=True,
                               stdout=subprocess.DEVNULL)
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
        elif _stale():
            _try_build(force=True)
        if not os.path.exists(EXT_PATH):
            raise _lib.MhaqFqError(
                f"{EXT_PATH} is missing: build it with `make -C mhaq_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`.  There is no Python fallback.")
        state = _stamp_state()
        if state == "stale":
            raise _lib.MhaqFqError(
                f"{EXT_PATH} was built for another torch or from other sources (stamp {STAMP_PATH} != "
                f"'{expected_stamp()}') and could not be rebuilt: run `make -B -C mhaq_amd/csrc _mhaq_torch.so`")
        if state == "unknown":
            # no stamp and no way to rebuild (a prebuilt extension on a machine without hipcc): its provenance is unknown,
            # not known-bad -- loaded with a warning; bind() below still refuses a C-ABI version mismatch
            print(f"[mhaq_amd] warning: {EXT_PATH} has no build stamp and could not be rebuilt; loading it unverified "
                  f"(expected stamp '{expected_stamp()}')", file=sys.stderr, flush=True)
        spec = importlib.util.spec_from_file_location("_mhaq_torch", EXT_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        built_for = getattr(mod, "TORCH_VERSION", None)
        running = torch.__version__.split("+")[0]
        if built_for is not None and str(built_for).split("+")[0] != running:
            raise _lib.MhaqFqError(
                f"{EXT_PATH} was compiled against torch {built_for}, this process runs torch {torch.__version__}: "
                f"run `make -B -C {os.path.dirname(EXT_PATH)} _mhaq_torch.so`")
        mod.bind(_lib.LIB_PATH, _lib.MhaqFqError)
        _ext = mod
    return _ext
