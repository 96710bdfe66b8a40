"""AdamW over the student's flat parameter store, with the construction rule of the reference
(src/training/main.py:198-213): group 0 = gain/bias-like tensors (ndim<2 or 'bn'/'ln'/'bias'/'logit_scale' in the
name) without weight decay, group 1 = the rest with `--wd`; betas/eps from `--beta1/--beta2/--eps`.

One HIP launch updates every tensor that received a gradient (fp32 master, moments, bf16 shadow); tensors whose
gradient is None in the reference (never reached by the dense path) are skipped entirely -- no decay either --
as torch.optim.AdamW does.  `param_groups`, `zero_grad`, `step`, `state_dict` follow the
torch.optim.Optimizer conventions so that the reference's scheduler / checkpoint code works unchanged.
"""
import math

import torch

from ..engine import is_no_decay


class FlatAdamW:
    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_divisor: float = 1.0):
        self.model = model
        self.engine = model.visual.engine
        self.grad_divisor = grad_divisor            # data-parallel world size (bucket all-reduce is SUM)
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        g0 = [p for n, p in named if is_no_decay(n, p.ndim)]
        g1 = [p for n, p in named if not is_no_decay(n, p.ndim)]
        common = dict(lr=lr, betas=tuple(betas), eps=eps, amsgrad=False, maximize=False)
        self.param_groups = [dict(params=g0, weight_decay=0.0, **common), dict(params=g1, weight_decay=weight_decay, **common)]
        self._engine_ids = {id(p) for p in model.visual._flat.values()}
        self._extra = [(n, p) for n, p in named if id(p) not in self._engine_ids]      # logit_scale
        self._extra_state = {}
        self.step_count = 0

    def zero_grad(self, set_to_none: bool = False):
        self.engine.zero_grad()
        for _, p in self._extra:
            p.grad = None

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        g0, g1 = self.param_groups
        b1, b2 = g1["betas"]
        self.engine.adamw_step(self.step_count, g1["lr"], g1["weight_decay"], b1, b2, g1["eps"],
                               grad_scale=1.0 / self.grad_divisor)
        for name, p in self._extra:                      # scalar stragglers (never have a grad in CLIPSelf)
            if p.grad is None:
                continue
            st = self._extra_state.setdefault(name, dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
            st["step"] += 1
            g = p.grad / self.grad_divisor
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = st["v"].sqrt() / math.sqrt(1 - b2 ** st["step"]) + g0["eps"]
            p.addcdiv_(st["m"], denom, value=-g0["lr"] / (1 - b1 ** st["step"]))

    # ---- torch.optim-compatible checkpoint format (src/training/main.py:304-309) ---------------------
    def _indexed(self):
        out, i = [], 0
        for grp in self.param_groups:
            for p in grp["params"]:
                out.append((i, p))
                i += 1
        return out

    def state_dict(self):
        by_id = {id(p): n for n, p in self.model.visual._flat.items()}
        eng, state = self.engine, {}
        active = (eng.flags & 1).cpu()
        for i, p in self._indexed():
            full = by_id.get(id(p))
            if full is None or self.step_count == 0:
                continue
            if not bool(active[eng.offsets[full][0] // 64]):
                continue
            state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=eng.view_of(eng.exp_avg, full).clone(),
                            exp_avg_sq=eng.view_of(eng.exp_avg_sq, full).clone())
        groups, i = [], 0
        for grp in self.param_groups:
            g = {k: v for k, v in grp.items() if k != "params"}
            g["params"] = list(range(i, i + len(grp["params"])))
            i += len(grp["params"])
            groups.append(g)
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        by_id = {id(p): n for n, p in self.model.visual._flat.items()}
        eng = self.engine
        for i, p in self._indexed():
            st = sd["state"].get(i)
            full = by_id.get(id(p))
            if st is None or full is None:
                continue
            eng.view_of(eng.exp_avg, full).copy_(st["exp_avg"].reshape(eng.logical[full]))
            eng.view_of(eng.exp_avg_sq, full).copy_(st["exp_avg_sq"].reshape(eng.logical[full]))
            self.step_count = int(st["step"])
        for grp, saved in zip(self.param_groups, sd["param_groups"]):
            grp.update({k: v for k, v in saved.items() if k != "params"})
