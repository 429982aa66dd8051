"""Drop-in for `pytorch_pretrained_bert.optimization` (BertAdam + warmup schedules) on MI355X.

`BertAdam.step()` (reference optimization.py:112-182) runs as ONE multi-tensor HIP launch pair
(vlp_bert_adam: per-tensor L2 clip :146-147, m/v update :151-152, update = m/(sqrt(v)+e) + wd*p :153-163,
scheduled lr :165-173, no bias correction :177-180) over flat fp32 master buffers instead of ~200 Python
iterations of ~8 torch kernels each.  Parameters with `grad is None` are skipped like the reference (:125-126).

Two parameter flavours are accepted:
  * fp16 parameters owned by a vlp_amd model (flat buffers of vlp_amd.engine.Engine): fp32 masters are kept
    inside the optimizer and the fp16 model copy is rewritten by the same kernel;
  * plain fp32 device tensors (any model): gradients are gathered into a flat fp32 buffer per step.
There is no CPU path.
"""
import math

import torch
from torch.optim import Optimizer
from torch.optim.optimizer import required

from . import _lib as K


def warmup_cosine(x, warmup=0.002):   # optimization.py:33-36
    if x < warmup:
        return x / warmup
    return 0.5 * (1.0 + math.cos(math.pi * x))


def warmup_constant(x, warmup=0.002):   # optimization.py:39-42
    if x < warmup:
        return x / warmup
    return 1.0


def warmup_linear(x, warmup=0.002):   # optimization.py:45-48
    if x < warmup:
        return x / warmup
    return max((x - 1.) / (warmup - 1.), 0)


SCHEDULES = {"warmup_cosine": warmup_cosine, "warmup_constant": warmup_constant, "warmup_linear": warmup_linear}


class _FlatGroup(object):
    """fp32 master / moment buffers of one param group plus the tensor boundary table."""

    def __init__(self, params):
        dev = params[0].device
        self.params = params
        offs = [0]
        for p in params:
            offs.append(offs[-1] + p.numel())
        self.n = offs[-1]
        self.offs = offs
        self.seg = torch.tensor(offs, device=dev, dtype=torch.int64)
        self.p32 = torch.cat([p.data.detach().float().reshape(-1) for p in params])
        self.m = torch.zeros_like(self.p32)
        self.v = torch.zeros_like(self.p32)
        self.norms = torch.zeros(K.bert_adam_norms_floats(self.n, len(params)), device=dev)    # per-tensor norms + per-chunk partials
        self.g32 = torch.zeros_like(self.p32)
        self.fp16 = params[0].dtype == torch.float16
        self.p16 = torch.empty(self.n, device=dev, dtype=torch.float16) if self.fp16 else None
        self.active = torch.ones(len(params), device=dev, dtype=torch.int32)


class BertAdam(Optimizer):
    """BERT Adam with weight-decay fix (same constructor as the reference, optimization.py:73)."""

    def __init__(self, params, lr=required, warmup=-1, t_total=-1, schedule="warmup_linear", b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.01, max_grad_norm=1.0):
        if lr is not required and lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        if not 0.0 <= b1 < 1.0:
            raise ValueError("Invalid b1 parameter: {} - should be in [0.0, 1.0[".format(b1))
        if not 0.0 <= b2 < 1.0:
            raise ValueError("Invalid b2 parameter: {} - should be in [0.0, 1.0[".format(b2))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e, weight_decay=weight_decay,
                        max_grad_norm=max_grad_norm)
        super(BertAdam, self).__init__(params, defaults)
        self._flat = None
        self._step = 0
        self.grad_scale = 1.0        # gradients arrive multiplied by this (a scaled fp16 backward); step() divides it out before the per-tensor clip

    def _engine(self):
        for g in self.param_groups:
            for p in g["params"]:
                return getattr(p, "_vlp_engine", None)
        return None

    def _build(self):
        eng = None
        for g in self.param_groups:
            for p in g["params"]:
                if not p.is_cuda:
                    raise RuntimeError("vlp_amd.BertAdam: parameters must live on the GPU (no CPU fallback)")
                e = getattr(p, "_vlp_engine", None)
                if e is None and hasattr(p, "_vlp_owner"):
                    e = p._vlp_owner
                eng = eng or e
        if eng is not None:
            eng.pack()
        self._flat = [_FlatGroup(list(g["params"])) for g in self.param_groups]
        # per-parameter state exactly as the reference keeps it (optimization.py:131-137): `next_m` / `next_v` are views of the flat
        # moment buffers, so `optimizer.state[p]` and the default state_dict() see the live values
        for fg in self._flat:
            for i, p in enumerate(fg.params):
                st = self.state[p]
                st["step"] = self._step
                st["next_m"] = fg.m[fg.offs[i]:fg.offs[i + 1]].view(p.shape)
                st["next_v"] = fg.v[fg.offs[i]:fg.offs[i + 1]].view(p.shape)

    def get_lr(self):   # optimization.py:96-110
        lr = []
        for group in self.param_groups:
            for _ in group["params"]:
                if self._step == 0:
                    return [0]
                if group["t_total"] != -1:
                    lr.append(group["lr"] * SCHEDULES[group["schedule"]](self._step / group["t_total"], group["warmup"]))
                else:
                    lr.append(group["lr"])
        return lr

    def state_dict(self):
        """torch.optim.Optimizer format with the reference's per-parameter entries (`step`, `next_m`, `next_v`; fp32 clones) plus
        `vlp_master_fp32` (the fp32 master weights of fp16 parameters -- not in the reference, which only runs BertAdam on fp32
        parameters, where the parameter IS the master)."""
        if self._flat is None:
            self._build()
        sd = super(BertAdam, self).state_dict()
        sd["state"] = {k: {kk: (vv.detach().clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd["state"].items()}
        sd["vlp_master_fp32"] = [fg.p32.detach().clone() for fg in self._flat]
        return sd

    def load_state_dict(self, sd):
        """Accepts this class's state_dict and a reference BertAdam checkpoint (per-parameter `next_m` / `next_v` / `step`, indexed in
        param_groups order).  Moments are restored in fp32 into the flat buffers; never silently dropped: a shape mismatch raises."""
        if self._flat is None:
            self._build()
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups) or any(len(a["params"]) != len(b["params"]) for a, b in zip(groups, self.param_groups)):
            raise ValueError("BertAdam.load_state_dict: parameter groups do not match the optimizer")
        step = 0
        for group, saved, fg in zip(self.param_groups, groups, self._flat):
            group.update({k: v for k, v in saved.items() if k != "params"})
            for i, pid in enumerate(saved["params"]):
                st = sd["state"].get(pid)
                if not st:
                    continue
                for key, buf in (("next_m", fg.m), ("next_v", fg.v)):
                    t = st[key]
                    if t.numel() != fg.offs[i + 1] - fg.offs[i]:
                        raise ValueError("BertAdam.load_state_dict: %s of parameter %s has %d elements, expected %d"
                                         % (key, pid, t.numel(), fg.offs[i + 1] - fg.offs[i]))
                    buf[fg.offs[i]:fg.offs[i + 1]].copy_(t.detach().reshape(-1).float())
                step = max(step, int(st.get("step", 0)))
        masters = sd.get("vlp_master_fp32")
        for j, fg in enumerate(self._flat):
            if masters is not None:
                fg.p32.copy_(masters[j])
            else:
                fg.p32.copy_(torch.cat([p.data.detach().float().reshape(-1) for p in fg.params]))
        self._step = step
        for fg in self._flat:
            for p in fg.params:
                self.state[p]["step"] = step

    def zero_grad(self, set_to_none=False):
        eng = self._engine()
        if eng is not None:
            eng.zero_grad()
            return
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    p.grad.detach_()
                    p.grad.zero_()

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._flat is None:
            self._build()
        eng = self._engine()
        unused = eng.unused_parameter_names() if eng is not None else set()
        name_of = {id(p): n for n, p in eng._params.items()} if eng is not None else {}
        for group, fg in zip(self.param_groups, self._flat):
            act = []
            for i, p in enumerate(fg.params):
                on = p.grad is not None and name_of.get(id(p)) not in unused
                act.append(1 if on else 0)
                if on:
                    fg.g32[fg.offs[i]:fg.offs[i + 1]].copy_(p.grad.detach().reshape(-1))
            if self.grad_scale != 1.0:
                fg.g32.mul_(1.0 / self.grad_scale)
            fg.active.copy_(torch.tensor(act, dtype=torch.int32), non_blocking=True)
            if group["t_total"] != -1:
                lr = group["lr"] * SCHEDULES[group["schedule"]](self._step / group["t_total"], group["warmup"])
            else:
                lr = group["lr"]
            K.bert_adam(fg.p32, fg.m, fg.v, fg.g32, True, fg.p16, fg.seg, len(fg.params), fg.n, fg.norms, lr=lr, b1=group["b1"],
                        b2=group["b2"], eps=group["e"], decay=group["weight_decay"], max_grad_norm=group["max_grad_norm"], active=fg.active)
            src = fg.p16 if fg.fp16 else fg.p32
            for i, p in enumerate(fg.params):
                if act[i]:
                    p.data.copy_(src[fg.offs[i]:fg.offs[i + 1]].view(p.shape))
        self._step += 1
        for group in self.param_groups:
            for p in group["params"]:
                self.state[p]["step"] = self._step
        return loss
