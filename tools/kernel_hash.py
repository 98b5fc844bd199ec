"""sha256 over the sources of the streaming kernels: ties a PMC traffic measurement (profiles/rNN_traffic.json)
to the code it was taken on.  Shared by bench.py, tools/summarize_prof.py and tests/test_profiles_cpu.py."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    h = hashlib.sha256()
    for f in ("fq_pt.hip", "fq_common.hpp"):
        with open(os.path.join(ROOT, "mhaq_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
