"""Teacher-forced quantizer check for model-level GPU tests.

A whole-model comparison (HIP layers vs the oracle's layers) is limited by the convolutions between the
quantizers, not by the quantizers: its tolerances say nothing about them.  This helper records, inside the
GPU model's own forward/backward, what every quantizer actually saw -- NoisyAct: input x and upstream gradient
g; NoisyConv2d / NoisyLinear: upstream gradient G of the quantized weight and, when the regulariser input
log2(max - min + s) took part in the loss, its gradient -- and then states each quantizer's gradients from
those tensors with the closed forms of oracle/fq_closed_form.py (fp64 sums; the device's own fp32 scale
bits).  The comparison is then the op-level bar: elementwise parts exact, reduced gradients within
1e-6 * sum|terms|.  Deterministic estimators only (LSQ): the random sign stream is not recorded."""
import math

import numpy as np
import torch

from oracle import fq_closed_form as CF
from tests.golden_util import exact_off_extremes

LN2 = math.log(2.0)


def _keep(rec, key):
    """Tensor hook that stores the gradient (a hook may be called with None when the tensor took no part in the
    loss, and must return None to leave the gradient alone)."""
    def hook(g):
        if g is not None:
            rec[key] = g.detach().clone()
    return hook


class Recorder:
    def __init__(self, model):
        self.acts, self.weights, self._handles = [], [], []
        for name, m in model.named_modules():
            if hasattr(m, "log_act_s") and hasattr(m, "act_b") and not getattr(m, "disable", False):
                rec = {"name": name, "m": m}
                self.acts.append(rec)
                self._handles.append(m.register_forward_pre_hook(self._act_pre(rec)))
                self._handles.append(m.register_forward_hook(self._act_post(rec)))
            elif hasattr(m, "log_wght_s") and hasattr(m, "weight"):
                rec = {"name": name, "m": m, "g_lwq": None}
                self.weights.append(rec)
                self._spy_weight(m, rec)

    @staticmethod
    def _act_pre(rec):
        def hook(mod, inp):
            rec["x"] = inp[0].detach().clone()
            if not inp[0].requires_grad:
                return None
            xv = inp[0].view_as(inp[0])       # a private alias: its gradient is this quantizer's gx alone,
            xv.register_hook(_keep(rec, "gx"))     # not the sum over a skip path
            return (xv,)
        return hook

    @staticmethod
    def _act_post(rec):
        def hook(mod, inp, out):
            out.register_hook(_keep(rec, "g"))
            rec["params"] = (mod.Q.scale.detach().clone(), mod.act_b.detach().clone(), mod.Q.max_val.detach().clone(),
                             torch.exp2(mod.log_act_q.detach()))
        return hook

    @staticmethod
    def _spy_weight(m, rec):
        inner = m._conv_forward if hasattr(m, "_conv_forward") else None
        if inner is None:
            raise NotImplementedError("Recorder handles NoisyConv2d layers")

        def spy(inp, weight, bias):
            rec["w"] = m.weight.detach().clone()
            rec["s"] = m.Q.scale.detach().clone()
            weight.register_hook(_keep(rec, "G"))
            lwq = getattr(m, "_lwq", None)
            if lwq is not None and lwq.requires_grad:      # (called with None when lwq took no part in the loss)
                lwq.register_hook(_keep(rec, "g_lwq"))
            return inner(inp, weight, bias)
        m._conv_forward = spy

    def close(self):
        for h in self._handles:
            h.remove()
        for rec in self.weights:
            rec["m"].__dict__.pop("_conv_forward", None)

    # ------------------------------------------------------------------ expectations
    @staticmethod
    def direct_grads(model, vals):
        """The part of each log-parameter's gradient that does NOT come through its quantizer: a regulariser
        (PotentialLoss) reads log_act_s / log_act_q / log_wght_s directly through the concatenated vectors of
        get_model_values.  `vals` = that call's (las, laq, lws, lwq) with retain_grad() set on the first three
        before backward; returns {id(parameter): gradient} in the order get_model_values concatenates them."""
        out = {}
        acts = [m for _, m in model.named_modules()
                if hasattr(m, "log_act_s") and hasattr(m, "log_act_q") and m.log_act_s.requires_grad]
        wls = [m for _, m in model.named_modules()
               if hasattr(m, "log_wght_s") and hasattr(m, "weight") and m.log_wght_s.requires_grad]
        las, laq, lws = (None if v.grad is None else v.grad.detach().reshape(-1) for v in vals[:3])
        for i, a in enumerate(acts):
            if las is not None:
                out[id(a.log_act_s)] = las[i:i + 1]
            if laq is not None:
                out[id(a.log_act_q)] = laq[i:i + 1]
        o = 0
        for m in wls:
            k = m.log_wght_s.numel()
            if lws is not None:
                out[id(m.log_wght_s)] = lws[o:o + k]
            o += k
        return out

    def check(self, rel=1e-6, direct=None):
        """Assert every recorded quantizer against its closed form; returns how many were checked.
        `direct`: direct_grads(...) of a loss that also reads the log-parameters themselves."""
        direct = direct or {}

        def through_op(p):
            g = p.grad.detach().double().reshape(-1).cpu()
            d = direct.get(id(p))
            return g if d is None else g - d.double().reshape(-1).cpu()
        n = 0
        for rec in self.acts:
            if "g" not in rec:
                continue
            m = rec["m"]
            s, b, hi, qr = (t.cpu() for t in rec["params"])
            x, g = rec["x"].cpu(), rec["g"].cpu()
            cf = CF.per_tensor(x, g, None, s, b, b, hi, "LSQ")
            sv, qv = float(s), float(qr)
            exp_ls = (float(cf["g_s"]) - float(cf["g_hi"])) * sv * LN2
            exp_lq = float(cf["g_hi"]) * qv * LN2
            exp_b = float(cf["g_zp"]) + float(cf["g_lo"]) + float(cf["g_hi"])
            abs_s, abs_g = float(cf["abs_s"]), float(cf["abs_g"])
            for p, exp, yard, what in ((m.log_act_s, exp_ls, (abs_s + abs_g) * sv * LN2, "log_act_s"),
                                       (m.log_act_q, exp_lq, abs_g * qv * LN2, "log_act_q"),
                                       (m.act_b, exp_b, abs_g, "act_b")):
                if p.grad is None:
                    assert not p.requires_grad, (rec["name"], what)
                    continue
                d = direct.get(id(p))
                slack = 0.0 if d is None else 1e-6 * (abs(float(d)) + abs(float(p.grad)))   # fp32 sum of the two parts
                err = abs(float(through_op(p)) - exp)
                assert err <= rel * yard + slack + 1e-30, (rec["name"], what, err, yard)
            if "gx" in rec:
                assert np.array_equal(rec["gx"].cpu().numpy(), cf["gx"].numpy()), (rec["name"], "gx")
            n += 1
        for rec in self.weights:
            if "G" not in rec:
                continue
            m = rec["m"]
            w, G = rec["w"].cpu(), rec["G"].cpu()
            pc = m.log_wght_s.numel() > 1
            co = w.shape[0] if pc else 1
            w2, G2 = w.reshape(co, -1), G.reshape(co, -1)
            s = rec["s"].cpu().reshape(co)
            cf = CF.per_channel(w2, G2, None, s, "LSQ")
            g_s = cf["g_s"].double().clone()
            gw = cf["gw"].double().clone()
            t = torch.zeros(co, dtype=torch.float64)
            if rec["g_lwq"] is not None:
                # lwq = log2((max - min) + s): d/d(max) = +t, d/d(min) = -t, d/ds = +t, t = g / (u ln2)
                mn, mx = w2.min(1).values, w2.max(1).values
                u = ((mx - mn) + s).double()
                t = rec["g_lwq"].cpu().reshape(co).double() / (u * LN2)
                g_s += t
                for c in range(co):
                    lo_m, hi_m = w2[c] == mn[c], w2[c] == mx[c]
                    gw[c][hi_m] += t[c] / int(hi_m.sum())
                    gw[c][lo_m] -= t[c] / int(lo_m.sum())
            exp_ls = (g_s * s.double() * LN2).numpy()
            yard = ((cf["abs_s"].double() + t.abs()) * s.double() * LN2).numpy()
            got = through_op(m.log_wght_s).numpy()
            slack = 0.0
            if id(m.log_wght_s) in direct:
                slack = 1e-6 * (np.abs(direct[id(m.log_wght_s)].cpu().double().numpy()) +
                                np.abs(m.log_wght_s.grad.detach().cpu().double().reshape(co).numpy()))
            err = np.abs(got - exp_ls)
            assert np.all(err <= rel * yard + slack + 1e-30), \
                (rec["name"], "log_wght_s", float(err.max()), float(yard.max()))
            gw_got = m.weight.grad.detach().cpu().reshape(co, -1).numpy()
            assert exact_off_extremes(gw_got, cf["gw"].numpy(), w2.numpy(), True, also_max=rec["g_lwq"] is not None), \
                (rec["name"], "gw off the row extremes")
            yard_g = (cf["abs_g"].double() + 2 * t.abs()).reshape(co, 1).numpy()
            errw = np.abs(gw_got.astype(np.float64) - gw.numpy())
            assert np.all(errw <= rel * (yard_g + np.abs(gw.numpy()))), (rec["name"], "gw", float(errw.max()))
            n += 1
        return n
