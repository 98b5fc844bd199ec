"""GPU (-m gpu): a torch-free C++ program links libmhaq_fq.so through include/mhaq_fq.h alone and checks the
activation forward/backward against a scalar host restatement, then the ABI v2 additions (split backward + joint
finalize, the device-resident stream offset, per-row min/max), the grouped weight backward for self-consistency and
the streaming per-tensor weight layer against a scalar host restatement, and the ABI v3 sign stream restated from the
header's text (tests/capi_smoke.cpp)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_native_consumer_of_the_c_abi(tmp_path):
    from mhaq_amd import _lib
    _lib.lib()
    exe = str(tmp_path / "capi_smoke")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-ffp-contract=off", "--offload-arch=gfx950",
                    os.path.join(HERE, "capi_smoke.cpp"), "-o", exe, f"-L{libdir}", "-lmhaq_fq",
                    f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    # the v1 checks, the ABI v2 additions, weight groups, the streaming per-tensor weight layer, the v3 sign stream
    assert out.stdout.count("-> OK") == 5, out.stdout
