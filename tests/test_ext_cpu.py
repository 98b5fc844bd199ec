"""CPU (-m "not gpu"): the compiled autograd binding (mhaq_amd/csrc/torch_binding.cpp -> _mhaq_torch.so) loads, binds the
C-ABI library mhaq_amd._lib uses, owns the sign-stream state behind mhaq_amd.ops.rng, and refuses host tensors -- no
compute calls without a GPU."""
import copy
import pickle

import pytest
import torch


def test_extension_loads_and_binds_the_same_library():
    from mhaq_amd import _ext, _lib
    E = _ext.ext()
    assert E.bound_library() == _lib.LIB_PATH
    for name in ("act_layer", "weight_layer", "weight_layer_pt", "potential_loss", "hub_create", "hub_begin",
                 "plan_create", "plan_forward", "plan_group_apply", "rng_next"):
        assert callable(getattr(E, name)), name


def test_sign_stream_state_is_one_object_for_python_and_compiled_ops():
    """ops.rng is a facade over the extension's generator: seeds are mixed with the rank, every draw is a fresh offset,
    the counter can be re-aligned (a capturing trainer does that around an eagerly run odd-shaped batch)."""
    from mhaq_amd import _ext, ops
    E = _ext.ext()
    ops.manual_seed(1234)
    assert ops.rng.seed == 1234 and ops.rng.drawn() == 0
    s1, o1 = ops.rng.next()
    s2, o2 = E.rng_next(0)
    assert (s1, o1) == (1234, 1) and (s2, o2) == (1234, 2) and ops.rng.drawn() == 2
    r1, o3 = E.rng_next(1)
    r2, _ = E.rng_next(2)
    assert o3 == 3 and len({s1, r1, r2}) == 3                       # ranks draw different streams
    assert r1 == 1234 ^ 0x9E3779B97F4A7C15
    ops.rng.set_drawn(1)
    assert ops.rng.next()[1] == 2
    assert ops.rng.offset_base is None
    with pytest.raises(ValueError):
        with ops.rng.device_offset(torch.zeros(1, dtype=torch.int64)):      # must live on the device
            pass


def test_compiled_ops_refuse_host_tensors():
    from mhaq_amd import _lib, ops
    x = torch.randn(2, 3, 4, 4)
    one = torch.zeros(1)
    with pytest.raises(_lib.MhaqFqError, match="no CPU fallback"):
        ops.fake_quant_act_layer(x, one, one, one, "STE")
    with pytest.raises(_lib.MhaqFqError, match="no CPU fallback"):
        ops.fake_quant_weight_layer(torch.randn(4, 3, 3, 3), torch.zeros(4, 1, 1, 1), "LSQ")
    with pytest.raises(_lib.MhaqFqError, match="no CPU fallback"):
        ops.fake_quant_weight_layer_pt(torch.randn(4, 3, 3, 3), one, "LSQ")
    with pytest.raises(AttributeError):
        ops.fake_quant_act_layer(x, one, one, one, "NOPE")
    with pytest.raises(NotImplementedError):
        ops.fake_quant_act_layer(x, one, one, one, "AEWGS")


def test_unknown_handles_raise_the_library_error():
    from mhaq_amd import _ext, _lib
    E = _ext.ext()
    with pytest.raises(_lib.MhaqFqError):
        E.hub_state(10 ** 9)
    with pytest.raises(_lib.MhaqFqError):
        E.plan_state(10 ** 9)


def test_hub_and_step_plumbing_do_not_travel_with_copies_or_pickles():
    """torch.save(model) / copy.deepcopy(model) carry parameters and buffers, not the activation hub's reference or a
    step's slab slices (`_hub`, `_pre_fwd`, `_lwq`, ...)."""
    import mhaq_amd as M
    from mhaq_amd.act_hub import ActGradHub
    net = torch.nn.Sequential(M.NoisyAct(), M.NoisyConv2d(3, 4, 3, qscheme=M.QScheme.PER_CHANNEL), M.NoisyAct())
    hub = ActGradHub(net)
    assert len(hub) == 2 and net[0]._hub.hub is hub
    net[1].__dict__["_pre_fwd"] = ("slab slices", None, None)
    net[1].__dict__["_lwq"] = torch.zeros(4)
    for clone in (copy.deepcopy(net), pickle.loads(pickle.dumps(net))):
        assert "_hub" not in clone[0].__dict__ and "_pre_fwd" not in clone[1].__dict__ and "_lwq" not in clone[1].__dict__
        assert torch.equal(clone[1].weight, net[1].weight)
        assert clone[1].regulariser_input() is None
    assert net[0]._hub.hub is hub                                   # the original keeps its hub
    with pytest.raises(TypeError):
        copy.deepcopy(hub)


def test_a_stale_extension_is_detected_by_its_build_stamp(tmp_path, monkeypatch):
    """The Makefile leaves "<torch version> <source hash>" next to _mhaq_torch.so; the loader compares it with the torch
    it runs under and the sources on disk BEFORE mapping the library -- an extension left over from a torch upgrade or an
    edited torch_binding.cpp is rebuilt (or refused), never loaded unchecked."""
    from mhaq_amd import _ext
    want = _ext.expected_stamp()
    assert want is not None and open(_ext.STAMP_PATH).read().strip() == want and not _ext._stale()
    version, src_hash = want.split()
    assert version == torch.__version__ and len(src_hash) == 16
    for text in (f"0.0.0 {src_hash}", f"{version} {'0' * 16}", ""):
        stamp = tmp_path / "stamp"
        stamp.write_text(text)
        monkeypatch.setattr(_ext, "STAMP_PATH", str(stamp))
        assert _ext._stale() and _ext._stamp_state() == "stale", text
    monkeypatch.setattr(_ext, "STAMP_PATH", str(tmp_path / "no_such_stamp"))
    # an extension without a stamp is rebuilt where that is possible too -- but its provenance is unknown, not known-bad:
    # where it cannot be rebuilt (prebuilt, no hipcc) it loads with a warning instead of being refused
    assert _ext._stale() and _ext._stamp_state() == "unknown"


def test_an_extension_without_a_stamp_that_cannot_be_rebuilt_loads_with_a_warning(tmp_path, monkeypatch, capfd):
    """A prebuilt _mhaq_torch.so without a .stamp on a machine where `make` fails (no hipcc) is loaded -- with a warning --
    and a stamp that names other sources is still refused.  (The loader's last step is stubbed: mapping the real library
    a second time into this process is not what is being tested.)"""
    import importlib.util
    import types
    from mhaq_amd import _ext, _lib
    calls, bound = [], []
    fake = types.SimpleNamespace(bind=lambda path, err: bound.append(path))
    loader = types.SimpleNamespace(exec_module=lambda mod: None)
    monkeypatch.setattr(importlib.util, "spec_from_file_location", lambda name, path: types.SimpleNamespace(loader=loader))
    monkeypatch.setattr(importlib.util, "module_from_spec", lambda spec: fake)
    monkeypatch.setattr(_ext, "_try_build", lambda force=False: calls.append(force))      # "the rebuild failed"
    monkeypatch.setattr(_ext, "_ext", None)
    monkeypatch.setattr(_ext, "STAMP_PATH", str(tmp_path / "no_such_stamp"))
    assert _ext.ext() is fake
    assert calls == [True] and bound == [_lib.LIB_PATH]
    assert "no build stamp" in capfd.readouterr().err
    stamp = tmp_path / "stamp"
    stamp.write_text("0.0.0 0000000000000000")
    monkeypatch.setattr(_ext, "_ext", None)
    monkeypatch.setattr(_ext, "STAMP_PATH", str(stamp))
    with pytest.raises(_lib.MhaqFqError, match="another torch or from other sources"):
        _ext.ext()
    assert calls == [True, True] and bound == [_lib.LIB_PATH]          # refused BEFORE anything is mapped
