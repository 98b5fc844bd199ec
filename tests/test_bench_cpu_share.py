"""CPU: bench.py's CPU legs size their thread pool to the box's CPU SHARE (tools/fq_sets.host_cores): the scheduler
affinity mask cut down to the cgroup quota -- a one-GPU box leases 16 CPUs of a 256-CPU host, and 256 threads against a
16-CPU quota ran one warm-up step of the CPU baseline for nine minutes."""
import builtins
import io
import os

from tools import fq_sets


def _with_files(monkeypatch, files, affinity):
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup"):
            if path in files:
                return io.StringIO(files[path])
            raise OSError(path)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(affinity)), raising=False)
    monkeypatch.delenv("MHAQ_CPU_THREADS", raising=False)


def test_cgroup_v2_quota_wins_over_the_affinity_mask(monkeypatch):
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}, 256)
    n, how = fq_sets.host_cores()
    assert n == 16 and "quota 16.0" in how and "256" in how


def test_cgroup_v1_quota(monkeypatch):
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "800000\n",
                              "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}, 64)
    assert fq_sets.host_cores()[0] == 8


def test_no_quota_small_mask_is_used_whole_and_a_large_one_is_capped_with_a_note(monkeypatch):
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu.max": "max 100000\n"}, 8)
    assert fq_sets.host_cores() == (8, "affinity mask 8")
    _with_files(monkeypatch, {}, 192)
    n, how = fq_sets.host_cores()
    assert n == 16 and "capped" in how


def test_explicit_override(monkeypatch):
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}, 256)
    monkeypatch.setenv("MHAQ_CPU_THREADS", "4")
    assert fq_sets.host_cores()[0] == 4
