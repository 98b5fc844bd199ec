"""CPU: host-side mirror of the reference interface -- constructor signatures, parameter
names / shapes / requires_grad, state_dict keys, enums, and loud failure without a GPU."""
import inspect
import os

import pytest
import torch

import mhaq_amd
from mhaq_amd import NoisyAct, NoisyConv2d, NoisyLinear, QNMethod, QScheme, Quantizer, _lib, ops


def test_enum_values_match_reference():
    # gdnsq_utils.py:9-13, aux/types.py:19-21
    assert [m.name for m in QNMethod] == ["STE", "EWGS", "AEWGS", "LSQ"]
    assert [m.value for m in QNMethod] == [0, 1, 2, 3]
    assert QScheme.PER_TENSOR.value == 0 and QScheme.PER_CHANNEL.value == 1


def test_constructor_signatures_match_reference():
    # gdnsq_act.py:10-18, gdnsq_conv2d.py:14-32, gdnsq_linear.py:14-25, gdnsq.py:160-169
    assert list(inspect.signature(NoisyAct.__init__).parameters)[1:] == [
        "init_s", "init_q", "signed", "noise_ratio", "disable", "qnmethod"]
    assert list(inspect.signature(NoisyConv2d.__init__).parameters)[1:] == [
        "in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation", "groups", "bias",
        "padding_mode", "device", "dtype", "qscheme", "log_s_init", "rand_noise", "quant_bias", "qnmethod"]
    assert list(inspect.signature(NoisyLinear.__init__).parameters)[1:] == [
        "in_features", "out_features", "bias", "device", "dtype", "qscheme", "log_s_init", "rand_noise",
        "qnmethod"]
    assert list(inspect.signature(Quantizer.__init__).parameters)[1:] == [
        "module", "scale", "zero_point", "min_val", "max_val", "rnoise_ratio", "qnmethod"]
    d = {k: v.default for k, v in inspect.signature(NoisyConv2d.__init__).parameters.items()}
    assert d["qscheme"] == QScheme.PER_TENSOR and d["log_s_init"] == -12 and d["qnmethod"] == QNMethod.AEWGS
    d = {k: v.default for k, v in inspect.signature(NoisyAct.__init__).parameters.items()}
    assert (d["init_s"], d["init_q"], d["signed"], d["qnmethod"]) == (-10, 10, True, QNMethod.STE)


def test_noisy_act_parameters():
    a = NoisyAct()
    sd = a.state_dict()
    assert set(sd) == {"log_act_q", "act_b", "log_act_s"}
    assert all(v.shape == (1,) for v in sd.values())
    assert float(a.log_act_s) == -10 and float(a.log_act_q) == 10 and float(a.act_b) == -512.0
    assert a.act_b.requires_grad and a.log_act_s.requires_grad and a.log_act_q.requires_grad
    u = NoisyAct(signed=False)
    assert float(u.act_b) == 0.0 and not u.act_b.requires_grad
    assert isinstance(a.Q, Quantizer) and a.bw.dim() == 0
    assert a.Q.qnmethod == QNMethod.STE and a.Q.positive_scale is True
    x = torch.randn(2, 3)
    assert NoisyAct(disable=True)(x) is x


def test_noisy_conv_parameters():
    c = NoisyConv2d(4, 8, 3, qscheme=QScheme.PER_CHANNEL)
    assert set(c.state_dict()) == {"weight", "bias", "log_wght_s", "log_b_s", "_noise_ratio"}
    assert c.log_wght_s.shape == (8, 1, 1, 1) and c.log_b_s.shape == (1,) and c._noise_ratio.shape == (1,)
    assert float(c.log_wght_s[0]) == -12 and not c._noise_ratio.requires_grad
    t = NoisyConv2d(4, 8, 3, bias=False)
    assert set(t.state_dict()) == {"weight", "log_wght_s", "_noise_ratio"} and t.log_wght_s.shape == (1,)
    assert t.Q.qnmethod == QNMethod.AEWGS
    with pytest.raises(AttributeError):       # like the reference: no log_b_s for PER_TENSOR
        NoisyConv2d(4, 8, 3, quant_bias=True)
    lin = NoisyLinear(16, 4, qscheme=QScheme.PER_CHANNEL)
    assert lin.log_wght_s.shape == (4, 1, 1, 1)
    assert "log_wght_s[8]" in repr(c) and "estimator=AEWGS" in repr(c) and "log_wght_s[4]" in repr(lin)


def test_product_path_fails_loudly_without_gpu():
    a = NoisyAct().train()
    with pytest.raises(_lib.MhaqFqError):
        a(torch.randn(2, 3, 4, 4))
    c = NoisyConv2d(3, 4, 3, qscheme=QScheme.PER_CHANNEL)
    with pytest.raises(_lib.MhaqFqError):
        c(torch.randn(1, 3, 8, 8))
    with pytest.raises(AttributeError):
        ops._method_value("NOPE")


def test_product_never_imports_oracle():
    import os
    import re
    root = os.path.dirname(os.path.abspath(mhaq_amd.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_bench_refuses_to_run_without_a_gpu():
    import subprocess
    import sys as _sys
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "no CPU fallback" in (out.stderr + out.stdout)


def test_weight_backward_groups_are_cut_from_the_end_of_the_model():
    """multi.backward_groups: the big late layers leave early (their all-reduce overlaps the rest of backward), the
    small early layers -- which DDP's last bucket holds back until the end anyway -- form the last group."""
    from mhaq_amd.multi import backward_groups
    r18 = [36864] * 4 + [73728] + [147456] * 3 + [294912] + [589824] * 3 + [1179648] + [2359296] * 3
    assert backward_groups(r18, [2] * 16, 4 << 20) == [(14, 16), (10, 14), (0, 10)]
    assert backward_groups(r18, [2] * 16, 1 << 40) == [(0, 16)]              # small models: one group
    assert backward_groups(r18, [2] * 16, 1) == []                           # every layer alone: per-layer launches
    # a group never mixes estimators; a layer left alone is not listed
    assert backward_groups([10, 10, 10, 10], [0, 0, 3, 0], 1000) == [(0, 2)]
    assert backward_groups([], [], 8) == []


def test_weight_backward_groups_properties():
    """For any layer sizes / estimators / threshold: groups are disjoint ranges in descending order, never mix
    estimators, hold at least two layers, and every group but the one nearest the front of its estimator run reaches the
    threshold; a layer outside every group is one the cut left alone."""
    from hypothesis import given, settings, strategies as st
    from mhaq_amd.multi import backward_groups

    @settings(max_examples=300, deadline=None)
    @given(st.lists(st.tuples(st.integers(1, 5000), st.integers(0, 3)), min_size=0, max_size=40), st.integers(1, 20000))
    def check(layers, min_elems):
        sizes = [a for a, _ in layers]
        methods = [b for _, b in layers]
        groups = backward_groups(sizes, methods, min_elems)
        prev_first = len(sizes)
        for first, last in groups:
            assert 0 <= first < last <= prev_first and last - first >= 2
            prev_first = first
            assert len({methods[i] for i in range(first, last)}) == 1
            total = sum(sizes[first:last])
            front_of_run = first == 0 or methods[first - 1] != methods[first]
            assert total >= min_elems or front_of_run
            # a group is closed as soon as it reaches the threshold: without its first layer it is still short
            assert sum(sizes[first + 1:last]) < min_elems
    check()


def test_ctypes_ops_make_their_inputs_device_current(monkeypatch):
    """include/mhaq_fq.h "Devices": the C ABI launches on the stream it is handed and wants that stream's device current.
    torch ops work on a tensor's device whatever the current one is (the reference is plain torch ops), so the ctypes entry
    points switch to the device of their first tensor / module / device argument for the duration of the call
    (ops._on_device; the compiled nodes hold a c10::OptionalDeviceGuard) -- and do nothing when it already is current."""
    import contextlib
    from mhaq_amd import ops
    entered = []

    @contextlib.contextmanager
    def fake_device(dev):
        entered.append(torch.device(dev))
        yield

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "device", fake_device)

    @ops._on_device
    def op(a, b=None, *rest):
        return "ran"

    class FakeTensor(torch.Tensor):            # a tensor that claims to live on cuda:1 (no GPU needed)
        @property
        def device(self):
            return torch.device("cuda:1")

    t1 = torch.zeros(2).as_subclass(FakeTensor)
    assert op(t1) == "ran" and entered == [torch.device("cuda:1")]
    assert op(5, 7, torch.device("cuda:1")) == "ran" and len(entered) == 2      # fill_r(n, seed, offset, device)
    assert op(3, "cuda:1") == "ran" and len(entered) == 3
    lin = torch.nn.Linear(2, 2)
    lin.weight.__class__ = type("P", (FakeTensor, torch.nn.Parameter), {})       # a module on cuda:1
    assert op(lin) == "ran" and len(entered) == 4
    # current device, host tensors, no tensor at all: straight through
    assert op(torch.device("cuda:0")) == "ran" and op(torch.zeros(1)) == "ran" and op(1, 2) == "ran" and len(entered) == 4
    # the FIRST tensor decides (the op's primary input), later arguments are not looked at
    assert op(torch.zeros(1), t1) == "ran" and len(entered) == 4
    # by keyword too: fake_quant_per_tensor(x=...), fill_r(n, seed, offset, device=...), device=<index>
    assert op(a=t1) == "ran" and len(entered) == 5
    assert op(5, b=torch.device("cuda:1")) == "ran" and len(entered) == 6

    @ops._on_device
    def filler(n, seed, offset, device=None):
        return "ran"
    assert filler(4, 1, 0, device="cuda:1") == "ran" and len(entered) == 7
    assert filler(4, 1, 0, device=1) == "ran" and entered[-1] == torch.device("cuda", 1) and len(entered) == 8
    assert filler(1, 1, 1, device=0) == "ran" and filler(1, 1, 1) == "ran" and len(entered) == 8     # an int that is not `device`
    for name in ("fill_r", "minmax", "row_minmax", "fake_quant_per_tensor", "fake_quant_per_tensor_eval",
                 "fake_quant_act_layer_eval", "fake_quant_weight_pc", "fake_quant_per_element", "fake_quant_weight_pt"):
        assert hasattr(getattr(ops, name), "__wrapped__"), name
