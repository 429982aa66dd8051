"""Drop-in for `pytorch_pretrained_bert.optimization_fp16` + the apex pieces it wraps, on MI355X.

The reference trains fp16 with `FP16_Optimizer_State(FusedAdam(groups, lr, bias_correction=False,
max_grad_norm=1.0), dynamic_loss_scale=True)` (run_img2txt_dist.py:411-420), `.backward(loss)` (:571),
per-step `param_group['lr'] = ...` (:580-583), `.step()` / `.zero_grad()` (:584-585).  apex is CUDA-only and
absent here; this module restates that contract on the flat buffers of vlp_amd.engine.Engine:

  per param group (weight-decay group / no-decay group):
      vlp_sumsq          -> grad L2 norm (still x loss scale) + inf/nan flag            [apex _compute_grad_norm]
      vlp_adam_hyper     -> clip folded into the unscale factor, skip flag             [FusedAdam.step scalar logic]
      vlp_fused_adam     -> fp32 master, m, v update + fp16 model copy in one pass     [fused_adam_cuda.adam]
  vlp_loss_scale_update  -> dynamic loss scale bookkeeping                               [FP16_Optimizer._update_scale]

Every scalar decision stays on the device: a training step issues no host synchronisation
(`cur_scale`, `overflow` are read back lazily, only when inspected).
"""
import logging
import os

import torch

from . import _lib as K
from .engine import is_no_decay

logger = logging.getLogger(__name__)


class FusedAdam(object):
    """Hyper-parameter holder with apex.optimizers.FusedAdam's constructor.  It only works wrapped in
    FP16_Optimizer_State (exactly how the reference uses it)."""

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-8, eps_inside_sqrt=False,
                 weight_decay=0., max_grad_norm=0., amsgrad=False):
        if amsgrad:
            raise RuntimeError("FusedAdam does not support the AMSGrad variant.")
        params = list(params)
        if len(params) == 0:
            raise ValueError("optimizer got an empty parameter list")
        if not isinstance(params[0], dict):
            params = [{"params": params}]
        self.defaults = dict(lr=lr, bias_correction=bias_correction, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        self.param_groups = []
        for g in params:
            g = dict(g)
            g["params"] = list(g["params"])
            for k, v in self.defaults.items():
                g.setdefault(k, v)
            self.param_groups.append(g)
        self.eps_mode = 0 if eps_inside_sqrt else 1
        self.state = {}

    def step(self, *a, **k):
        raise NotImplementedError("vlp_amd.FusedAdam is driven by FP16_Optimizer_State.step() (the only way the reference uses it)")

    def zero_grad(self):
        pass

    def state_dict(self):
        return {"param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups], "state": self.state}

    def load_state_dict(self, sd):
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
        self.state = sd.get("state", {})


class FP16_Optimizer_State(object):
    """apex FP16_Optimizer (FusedAdam flavour) + the state_dict extensions of the reference's subclass
    (optimization_fp16.py:17-80)."""

    def __init__(self, init_optimizer, static_loss_scale=1.0, dynamic_loss_scale=False, dynamic_loss_args=None, verbose=True):
        if not isinstance(init_optimizer, FusedAdam):
            raise TypeError("FP16_Optimizer_State expects a vlp_amd.optimization_fp16.FusedAdam")
        self.optimizer = init_optimizer
        self.param_groups = init_optimizer.param_groups
        eng = None
        for g in self.param_groups:
            for p in g["params"]:
                eng = eng or getattr(p, "_vlp_engine", None) or getattr(p, "_vlp_owner", None)
        if eng is None:
            raise RuntimeError("FP16_Optimizer_State: parameters do not belong to a vlp_amd model")
        eng.pack()
        self.engine = eng
        # map the caller's param groups onto the engine's two flat buffers
        name_of = {id(p): n for n, p in eng._params.items()}
        self._group_key = []
        for g in self.param_groups:
            names = [name_of[id(p)] for p in g["params"]]
            kinds = {is_no_decay(n) for n in names}
            if len(kinds) != 1:
                raise RuntimeError("FP16_Optimizer_State: a param group mixes decay / no-decay parameters; use the reference's two groups "
                                   "(run_img2txt_dist.py:394-401)")
            key = "nodecay" if kinds.pop() else "decay"
            if set(names) != set(eng.names[key]):
                raise RuntimeError("FP16_Optimizer_State: group does not cover the model's whole %s set" % key)
            self._group_key.append(key)
        if sorted(self._group_key) != ["decay", "nodecay"]:
            raise RuntimeError("FP16_Optimizer_State: expected exactly one decay and one no-decay group")
        dev = eng.device
        self.fp32_groups_flat = [eng.flat[k].float() for k in self._group_key]      # master weights
        self._m = [torch.zeros_like(t) for t in self.fp32_groups_flat]
        self._v = [torch.zeros_like(t) for t in self.fp32_groups_flat]
        self._sumsq = [torch.zeros(2, device=dev) for _ in self._group_key]
        self._hyper = [torch.zeros(3, device=dev) for _ in self._group_key]
        self._partial = torch.zeros(2048, device=dev)
        self._ovf = torch.zeros(1, device=dev)
        self.dynamic_loss_scale = bool(dynamic_loss_scale)
        if dynamic_loss_scale:
            args = dynamic_loss_args or {}
            init, factor, window = args.get("init_scale", 2 ** 16), args.get("scale_factor", 2), args.get("scale_window", 1000)
        else:
            init, factor, window = static_loss_scale, 2, 1000
        # {cur_scale, cur_iter, last_overflow_iter, scale_factor, scale_window, dynamic, skipped, -}
        self._scale_state = torch.tensor([init, 0, -1, factor, window, 1.0 if dynamic_loss_scale else 0.0, 0, 0], device=dev, dtype=torch.float32)
        self.verbose = verbose
        # True: step() runs on a second stream, chunked in the order the next forward reads the parameters (see _step_pipelined).  Set by
        # the train loops that only touch parameters through the engine (vlp_amd.run_img2txt_dist, bench.py); default off because code
        # that reads parameter storage directly right after step() would have to call engine.wait_params() first.
        self.pipeline_with_forward = os.environ.get("VLP_ADAM_PIPELINE", "0") == "1"
        # Adam's step count = APPLIED steps only (apex increments state['step'] inside the update, which an overflow skips).  The
        # skip decision lives on the device, so the count is derived from the device-side counters when it is needed:
        #   applied = _applied0 + (cur_iter - _iter0) - (skipped - _skipped0)
        self._applied0, self._iter0, self._skipped0 = 0, 0, 0
        self._steps_issued = 0                    # host-side count of step() calls (no device read-back)
        self._state_gathered_at = -1              # value of _steps_issued at the last consolidate() of a sharded optimizer

    # ---- lazily synchronised views of the device-side state -------------------------------------------
    def _sync(self):
        self.engine.wait_params(host=True)

    @property
    def cur_scale(self):
        self._sync()
        return float(self._scale_state[0])

    @property
    def cur_iter(self):
        self._sync()
        return int(self._scale_state[1])

    @property
    def last_overflow_iter(self):
        self._sync()
        return int(self._scale_state[2])

    @property
    def scale_factor(self):
        self._sync()
        return float(self._scale_state[3])

    @property
    def scale_window(self):
        self._sync()
        return int(self._scale_state[4])

    @property
    def overflow(self):
        self._sync()
        return bool(self._ovf[0] != 0)

    @property
    def skipped_steps(self):
        self._sync()
        return int(self._scale_state[6])

    @property
    def applied_steps(self):
        """Number of optimizer updates that were really applied (overflow steps excluded); one host read-back."""
        self._sync()
        st = self._scale_state.tolist()
        return int(self._applied0 + (st[1] - self._iter0) - (st[6] - self._skipped0))

    # ---- the train-loop contract -------------------------------------------------------------------------
    def backward(self, loss):
        """apex: scaled_loss = loss.float() * cur_scale; scaled_loss.backward()   (run_img2txt_dist.py:571)"""
        # a pipelined step (pipeline_with_forward) may still be running on the optimizer stream: its loss_scale_update is the LAST thing it
        # enqueues, so the scale read here must be ordered behind the whole step, not only behind the first parameter chunk
        self.engine.wait_params()
        # d(loss * scale) = scale * d(loss): the device-resident scale goes in as the upstream gradient instead of being multiplied into the
        # loss first -- the same numbers without the multiply, its autograd twin and the ones() fill (three launches between forward and backward)
        # NOTE: the upstream gradient below is a live VIEW of the optimizer's scale state -- _LossFn.backward hands its pointer to the
        # backward kernels without a copy.  That is correct because every writer of _scale_state (vlp_loss_scale_update, last launch of a step) is
        # ordered in front of this point: wait_params() above joins the optimizer stream, and the next update is only enqueued by step(), after
        # this backward.  Anyone who moves loss_scale_update to another stream has to clone the element here instead.
        loss = loss.float()
        loss.backward(gradient=self._scale_state[0:1].reshape(loss.shape))

    def zero_grad(self, set_grads_to_None=True):
        self.engine.zero_grad()

    def _step_sharded(self, plan):
        """The optimizer step sharded over the data-parallel ranks (vlp_amd.distributed.ShardPlan, VLP_DDP_MODE=sharded).  After the
        reduce-scatter rank r holds the mean gradient of chunk r of every bucket: it squares-and-sums those chunks, the per-group
        (sum, overflow) pairs of all ranks meet in ONE 4-float all-reduce (so every rank derives the same clip factor and the same
        skip decision), it runs the fused Adam kernel on its chunks of master / m / v only, and the updated fp16 parameters are
        all-gathered bucket by bucket (the bytes of the gradient all-gather this replaces); the next forward waits per bucket.  Per
        element the arithmetic is the unsharded kernel's."""
        eng = self.engine
        eng.wait_params()
        for i, key in enumerate(self._group_key):
            first = True
            for lo, hi in plan.owned(key):
                K.sumsq(eng.gflat[key][lo:hi], hi - lo, self._sumsq[i], self._partial, accumulate=not first)
                first = False
        stats = torch.cat(self._sumsq)                      # [sum0, flag0, sum1, flag1]
        plan.exchange_norms(stats)
        for i in range(len(self._group_key)):
            self._sumsq[i].copy_(stats[2 * i:2 * i + 2])
        torch.maximum(self._sumsq[0][1:2], self._sumsq[1][1:2], out=self._ovf)
        for i, key in enumerate(self._group_key):
            g = self.param_groups[i]
            K.adam_hyper(self._sumsq[i], self._ovf, self._scale_state, g["max_grad_norm"], self._step_size(g), self._hyper[i])
            for lo, hi in plan.owned(key):
                self._adam_range(i, key, lo, hi)
        K.loss_scale_update(self._scale_state, self._ovf)
        eng._param_works = plan.gather_params(eng.flat["decay"], eng.flat["nodecay"])

    def consolidate(self):
        """COLLECTIVE (VLP_DDP_MODE=sharded): every rank calls it before ANY rank asks for state_dict() / apex_state_dict() -- master /
        m / v are current only on a rank's own chunks and are all-gathered here.  A train loop that checkpoints on rank 0 only calls
        consolidate() on all ranks first (vlp_amd/run_img2txt_dist.py); state_dict() on an unconsolidated sharded optimizer gathers
        itself, which is correct only when every rank calls it.  No-op for the replicated step."""
        self._gather_sharded_state()

    def _gather_sharded_state(self):
        """Before a checkpoint: master / m / v are current only on this rank's chunks."""
        plan = getattr(self.engine, "shard_plan", None)
        if plan is None or self._state_gathered_at == self._steps_issued:
            return
        self._state_gathered_at = self._steps_issued
        self.engine.wait_params()
        i_d, i_nd = self._group_key.index("decay"), self._group_key.index("nodecay")
        plan.gather_state([self.fp32_groups_flat[i_d], self._m[i_d], self._v[i_d]], [self.fp32_groups_flat[i_nd], self._m[i_nd], self._v[i_nd]])

    def step(self, closure=None):
        eng = self.engine
        self._steps_issued += 1
        plan = getattr(eng, "shard_plan", None)
        if plan is not None:
            return self._step_sharded(plan)
        if self.pipeline_with_forward:
            return self._step_pipelined()
        eng.wait_params()                        # a previous pipelined step may still be writing
        self._grad_norms()
        # apex skips the whole step when ANY group overflowed
        torch.maximum(self._sumsq[0][1:2], self._sumsq[1][1:2], out=self._ovf)
        for i, key in enumerate(self._group_key):
            g = self.param_groups[i]
            K.adam_hyper(self._sumsq[i], self._ovf, self._scale_state, g["max_grad_norm"], self._step_size(g), self._hyper[i])
            self._adam_range(i, key, 0, eng.sizes[key])
        K.loss_scale_update(self._scale_state, self._ovf)

    def _grad_norms(self):
        """(sum of squares, overflow flag) of every param group's gradient -> self._sumsq[i]: apex FP16_Optimizer's overflow check + norm."""
        eng = self.engine
        for i, key in enumerate(self._group_key):
            K.sumsq(eng.gflat[key], eng.sizes[key], self._sumsq[i], self._partial)

    def _step_size(self, g):
        if g["bias_correction"]:
            b1, b2 = g["betas"]
            t = self.applied_steps + 1              # not the reference's configuration (bias_correction=False): costs a host sync
            return g["lr"] * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t)
        return g["lr"]

    def _adam_range(self, i, key, lo, hi):
        g = self.param_groups[i]
        b1, b2 = g["betas"]
        eng = self.engine
        K.fused_adam(self.fp32_groups_flat[i][lo:hi], self._m[i][lo:hi], self._v[i][lo:hi], eng.gflat[key][lo:hi], eng.flat[key][lo:hi], hi - lo,
                     self._hyper[i], b1=b1, b2=b2, eps=g["eps"], decay=g["weight_decay"], eps_inside_sqrt=(self.optimizer.eps_mode == 0))

    def _step_pipelined(self):
        """The same update, issued on the engine's optimizer stream in the order the NEXT forward consumes the parameters (no-decay
        group, embeddings + region projections, layer 0 ... layer N-1, task head), one event per chunk: the next forward waits for
        the chunk it is about to read instead of for the whole 3.25 GB pass, so the HBM-bound update runs underneath the (MFMA /
        latency-bound) GEMMs of the first layers.  Same kernels, same per-element arithmetic: results are bit-identical to step().
        The gradient norm (hence the clip factor and the overflow decision) still needs ALL gradients first."""
        eng = self.engine
        main = torch.cuda.current_stream()
        st = eng.optimizer_stream()
        eng.wait_params()                        # chunks of the previous step (normally long done)
        st.wait_stream(main)                     # backward (and the gradient all-reduce) is complete in main's order
        events = {}
        with torch.cuda.stream(st):
            self._grad_norms()
            torch.maximum(self._sumsq[0][1:2], self._sumsq[1][1:2], out=self._ovf)
            for i, key in enumerate(self._group_key):
                g = self.param_groups[i]
                K.adam_hyper(self._sumsq[i], self._ovf, self._scale_state, g["max_grad_norm"], self._step_size(g), self._hyper[i])
            i_nd, i_d = self._group_key.index("nodecay"), self._group_key.index("decay")
            self._adam_range(i_nd, "nodecay", 0, eng.sizes["nodecay"])
            ev = torch.cuda.Event()
            ev.record(st)
            events["nodecay"] = ev
            for b in reversed(range(len(eng.buckets))):          # slice len-1 = embeddings + region projections, ..., slice 0 = head
                lo, hi = eng.buckets[b]
                self._adam_range(i_d, "decay", lo, hi)
                ev = torch.cuda.Event()
                ev.record(st)
                events[b] = ev
            K.loss_scale_update(self._scale_state, self._ovf)
            done = torch.cuda.Event()
            done.record(st)
        eng.set_param_events(events, done)

    # ---- checkpointing (optimization_fp16.py:17-80) ----------------------------------------------------------
    def state_dict(self):
        self._gather_sharded_state()
        self._sync()
        sd = {"dynamic_loss_scale": self.dynamic_loss_scale, "cur_scale": self.cur_scale, "cur_iter": self.cur_iter}
        if self.dynamic_loss_scale:
            sd.update(last_overflow_iter=self.last_overflow_iter, scale_factor=self.scale_factor, scale_window=self.scale_window)
        inner = self.optimizer.state_dict()
        inner["exp_avg"] = [t.clone() for t in self._m]
        inner["exp_avg_sq"] = [t.clone() for t in self._v]
        inner["step"] = self.applied_steps
        inner["group_keys"] = list(self._group_key)
        sd["optimizer_state_dict"] = inner
        sd["fp32_groups_flat"] = [t.clone() for t in self.fp32_groups_flat]
        # not in the reference's format: the engine's dropout stream position, so that a resumed run continues the SAME mask sequence
        sd["vlp_rng"] = {"base_seed": self.engine.base_seed, "step_seed": self.engine.step_seed}
        return sd

    # ---- interchange with the reference stack's files (apex FP16_Optimizer layout) ------------------------------------------
    def _group_layout(self, i):
        """[(engine offset, numel)] of the parameters of param group i in the CALLER'S order -- apex flattens a group's parameters densely in
        exactly that order (apex/optimizers/fp16_optimizer.py: _flatten_dense_tensors over param_group['params']), whereas the engine keeps
        them in backward-completion order on 128-byte boundaries."""
        eng = self.engine
        name_of = {id(p): n for n, p in eng._params.items()}
        offs = eng.offsets[self._group_key[i]]
        return [(offs[name_of[id(p)]], p.numel()) for p in self.param_groups[i]["params"]]

    def _to_dense(self, flat, i):
        return torch.cat([flat[o:o + n] for o, n in self._group_layout(i)])

    def _from_dense(self, dense, flat, i):
        lay = self._group_layout(i)
        if dense.numel() != sum(n for _, n in lay):
            raise ValueError("apex state: group %d holds %d elements, this model's group has %d" % (i, dense.numel(), sum(n for _, n in lay)))
        pos = 0
        for o, n in lay:
            flat[o:o + n].copy_(dense[pos:pos + n])
            pos += n

    def apex_state_dict(self):
        """The checkpoint the REFERENCE stack writes (optimization_fp16.py:17-38 on apex's FP16_Optimizer + FusedAdam): the inner optimizer
        is a torch Optimizer whose param groups each hold ONE parameter -- the group's flat fp32 master -- so
        optimizer_state_dict = {state: {gid: {step, exp_avg, exp_avg_sq}}, param_groups: [{..., params: [gid]}]} with dense flat tensors in
        the caller's parameter order, and fp32_groups_flat likewise.  An optim.N.bin saved from this dict loads into the reference."""
        self._gather_sharded_state()
        self._sync()
        sd = {"dynamic_loss_scale": self.dynamic_loss_scale, "cur_scale": self.cur_scale, "cur_iter": self.cur_iter}
        if self.dynamic_loss_scale:
            sd.update(last_overflow_iter=self.last_overflow_iter, scale_factor=self.scale_factor, scale_window=self.scale_window)
        step = self.applied_steps
        state, groups = {}, []
        for i, g in enumerate(self.param_groups):
            state[i] = {"step": step, "exp_avg": self._to_dense(self._m[i], i), "exp_avg_sq": self._to_dense(self._v[i], i)}
            hp = {k: v for k, v in g.items() if k != "params"}
            hp["params"] = [i]
            groups.append(hp)
        sd["optimizer_state_dict"] = {"state": state, "param_groups": groups}
        sd["fp32_groups_flat"] = [self._to_dense(self.fp32_groups_flat[i], i) for i in range(len(self.param_groups))]
        return sd

    def load_apex_state_dict(self, sd):
        """Resume from an optim.N.bin written by the reference stack (same two param groups in the same parameter order,
        run_img2txt_dist.py:394-401)."""
        self._sync()
        inner = sd["optimizer_state_dict"]
        if len(inner["param_groups"]) != len(self.param_groups):
            raise ValueError("apex state: %d param groups, expected %d" % (len(inner["param_groups"]), len(self.param_groups)))
        self.dynamic_loss_scale = sd["dynamic_loss_scale"]
        st = self._scale_state.cpu()
        st[0], st[1] = sd["cur_scale"], sd["cur_iter"]
        if sd["dynamic_loss_scale"]:
            st[2], st[3], st[4], st[5] = sd["last_overflow_iter"], sd["scale_factor"], sd["scale_window"], 1.0
        self._scale_state.copy_(st)
        step = 0
        for i, (g, saved) in enumerate(zip(self.param_groups, inner["param_groups"])):
            g.update({k: v for k, v in saved.items() if k != "params"})
            pid = saved["params"][0]
            stt = inner["state"].get(pid) or inner["state"].get(str(pid)) or {}
            if stt:
                self._from_dense(stt["exp_avg"].to(self._m[i].device).float().reshape(-1), self._m[i], i)
                self._from_dense(stt["exp_avg_sq"].to(self._v[i].device).float().reshape(-1), self._v[i], i)
                step = max(step, int(stt.get("step", 0)))
            self._from_dense(sd["fp32_groups_flat"][i].detach().to(self._m[i].device).float().reshape(-1), self.fp32_groups_flat[i], i)
        stl = self._scale_state.tolist()
        self._applied0, self._iter0, self._skipped0 = step, stl[1], stl[6]
        for i, key in enumerate(self._group_key):       # refresh the fp16 model copy from the restored masters
            self.engine.flat[key].copy_(self.fp32_groups_flat[i])
        # apex's layout has no slot for the engine's dropout stream position: weights, moments and the loss scale resume exactly, the
        # dropout MASK sequence restarts from this process's seed (a native optim.N.bin carries `vlp_rng` and resumes mask-exactly)
        logger.info("FP16_Optimizer_State: resumed from an apex-layout file; the dropout stream position is not part of that layout "
                    "(masks restart from base_seed=%d, step_seed=%d)", self.engine.base_seed, self.engine.step_seed)

    def load_state_dict(self, sd):
        inner = sd.get("optimizer_state_dict", {})
        # the reference stack's layout is recognised by its STRUCTURE (a torch Optimizer state_dict: 'state' + 'param_groups', none of
        # this class's own keys) -- also when the inner state is still empty (a file saved before the first applied step)
        if "state" in inner and "param_groups" in inner and "exp_avg" not in inner and "group_keys" not in inner:
            return self.load_apex_state_dict(sd)
        self._sync()
        self.dynamic_loss_scale = sd["dynamic_loss_scale"]
        st = self._scale_state.cpu()
        st[0], st[1] = sd["cur_scale"], sd["cur_iter"]
        if sd["dynamic_loss_scale"]:
            st[2], st[3], st[4], st[5] = sd["last_overflow_iter"], sd["scale_factor"], sd["scale_window"], 1.0
        self._scale_state.copy_(st)
        inner = sd["optimizer_state_dict"]
        self.optimizer.load_state_dict(inner)
        for cur, saved in zip(self._m, inner["exp_avg"]):
            cur.copy_(saved)
        for cur, saved in zip(self._v, inner["exp_avg_sq"]):
            cur.copy_(saved)
        st = self._scale_state.tolist()
        self._applied0, self._iter0, self._skipped0 = int(inner.get("step", 0)), st[1], st[6]
        for cur, saved in zip(self.fp32_groups_flat, sd["fp32_groups_flat"]):
            cur.data.copy_(saved.data)
        for i, key in enumerate(self._group_key):       # refresh the fp16 model copy from the restored masters
            self.engine.flat[key].copy_(self.fp32_groups_flat[i])
        if "vlp_rng" in sd:
            self.engine.base_seed, self.engine.step_seed = int(sd["vlp_rng"]["base_seed"]), int(sd["vlp_rng"]["step_seed"])
