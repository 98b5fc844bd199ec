"""Shared by the CPU and GPU calibration tests: rebuild the flat layer lists of tests/golden/calib_cases.npz
(oracle/gen_golden.py::gen_calib) from any layer classes and compare the calibrated state with the reference's."""
import torch

from tests.golden_util import T, bit_equal

N_ACT, N_CONV, N_BATCH = 6, 4, 3
SIGNED = (True, False, True, True, True, False)
CONV_SHAPES = ((3, 6, 3), (6, 4, 3), (4, 5, 1), (5, 8, 3))


class Feed(torch.nn.Module):
    """forward(b): every quantizer sees its recorded batch b (the observer only needs the inputs)."""

    def __init__(self, acts, case, device):
        super().__init__()
        self.acts = acts
        self.x = [[T(case[f"act{i}_x{b}"]).to(device) for b in range(N_BATCH)] for i in range(N_ACT)]

    def forward(self, b):
        for i, a in enumerate(self.acts):
            a(self.x[i][int(b)])
        return None


def build(case, Act, Conv, per_channel, device):
    acts = torch.nn.ModuleList([Act(signed=s) for s in SIGNED])
    for i, a in enumerate(acts):
        gs, gq, gb = (bool(v) for v in case[f"act{i}_grad_in"])
        a.log_act_s.requires_grad_(gs), a.log_act_q.requires_grad_(gq), a.act_b.requires_grad_(gb)
    convs = torch.nn.ModuleList([Conv(ci, co, k, qscheme=per_channel) for ci, co, k in CONV_SHAPES])
    with torch.no_grad():
        for i, c in enumerate(convs):
            c.weight.copy_(T(case[f"conv{i}_w"]))
            c.log_wght_s.copy_(T(case[f"conv{i}_log_wght_s_in"]).view_as(c.log_wght_s))
            c.log_wght_s.requires_grad_(bool(case[f"conv{i}_grad_in"]))
    return acts.to(device), convs.to(device)


def assert_matches_reference(case, acts, convs):
    for i, a in enumerate(acts):
        for name in ("log_act_s", "log_act_q", "act_b"):
            got = getattr(a, name).detach().cpu().float().numpy().reshape(-1)
            assert bit_equal(got, case[f"act{i}_{name}"].reshape(-1)), (i, name, got, case[f"act{i}_{name}"])
        flags = [a.log_act_s.requires_grad, a.log_act_q.requires_grad, a.act_b.requires_grad]
        assert flags == [bool(v) for v in case[f"act{i}_grad_out"]], (i, flags)
    for i, c in enumerate(convs):
        got = c.log_wght_s.detach().cpu().numpy()
        assert bit_equal(got.reshape(-1), case[f"conv{i}_log_wght_s"].reshape(-1)), (i, got.ravel())
        assert c.log_wght_s.requires_grad == bool(case[f"conv{i}_grad_out"])


def oracle_states(case):
    """The same fixtures as plain records for oracle/calib.py."""
    acts = []
    for i in range(N_ACT):
        xs = [T(case[f"act{i}_x{b}"]) for b in range(N_BATCH)]
        gs, gq, gb = (bool(v) for v in case[f"act{i}_grad_in"])
        acts.append(dict(batches=xs, grad_s=gs, grad_q=gq, grad_b=gb))
    convs = [dict(weight=T(case[f"conv{i}_w"]), log_wght_s=T(case[f"conv{i}_log_wght_s_in"]),
                  grad=bool(case[f"conv{i}_grad_in"])) for i in range(N_CONV)]
    return acts, convs
