"""CPU: the HBM-traffic figure bench.py reports (`roofline.traffic`) must come from a PMC measurement of the
CURRENT kernels: profiles/rNN_traffic.json (tools/summarize_prof.py, from the two rocprofv3 --pmc passes of
tools/profile_bench.sh) records a hash of the streaming-kernel sources, and this test fails when they have
changed since -- re-run the passes instead of carrying a stale constant."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_newest_traffic_profile_matches_the_kernel_sources():
    import sys
    sys.path.insert(0, ROOT)
    from tools.kernel_hash import kernel_source_hash
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    assert files, "no profiles/rNN_traffic.json: run tools/profile_bench.sh on the GPU box and copy traffic.json"
    rec = json.load(open(files[-1]))
    assert rec["kernel_source_hash"] == kernel_source_hash(), \
        f"{os.path.basename(files[-1])} was measured on other kernel sources: re-run tools/profile_bench.sh"
    ent = rec["kernels"]["mhaq::pt_bwd_kernel<0, false, true, false, true, true>"]
    alg = 12 * ent["tensor_elements"]
    # no wasted re-reads: HBM traffic within 3 % of the algorithmic 12 B/elem
    assert 0.97 * alg <= ent["hbm_bytes"] <= 1.03 * alg, (ent["hbm_bytes"], alg)
