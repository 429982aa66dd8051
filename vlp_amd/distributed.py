"""Data-parallel gradient reduction for the fused engine: RCCL over xGMI, one process per GPU.

Replaces `torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
find_unused_parameters=True)` of the reference (run_img2txt_dist.py:379-386).  torch's DDP hangs its reducer on
autograd's AccumulateGrad hooks; the fused engine writes gradients into flat buffers without a per-parameter
autograd graph, so the reducer lives here instead:

  * construction broadcasts rank 0's flat parameter buffers (DDP's initial parameter sync);
  * the engine lays parameters out in backward-completion order, so every gradient bucket is a CONTIGUOUS slice
    of the flat fp16 gradient buffer -- no flatten/copy.  As soon as the last weight-gradient GEMM of a bucket has
    been launched the engine calls `bucket_ready(i)` and the slice goes to `all_reduce(async_op=True)`: RCCL runs
    it on its own stream, ordered after the producing kernels, overlapped with the rest of backward;
  * buckets are coalesced up to `bucket_cap_mb` (default 50 MB = 3-4 BERT-base layers): xGMI is point-to-point
    (7 links x ~153 GB/s), so fewer, larger collectives beat many small ones;
  * the reference's unused parameters (pooler; LM head in VQA) are static, so there is no per-step
    unused-parameter bitmap exchange: their gradient slots stay zero;
  * reduction is the MEAN over ranks (torch DDP semantics) -- ReduceOp.AVG on RCCL; on
    backends without AVG (gloo, used by the CPU tests) SUM followed by a 1/world scale;
  * exchange form: one all-reduce per bucket, or reduce-scatter + all-gather per bucket (`rs_ag`; SURVEY.md section 5 / 8e argues that on
    point-to-point xGMI the direct reduce-scatter / all-gather uses all 7 links of a GPU while a ring is bound by one: 0.38 vs 2.65 ms for
    231.9 MB); same result up to fp16 summation order.  THE RULE (round 6, `choose_mode`): `VLP_DDP_MODE` if set; otherwise all-reduce below
    4 ranks (two or three GPUs: a ring over the direct link IS the direct exchange), and from 4 ranks on RCCL the form is MEASURED once at
    construction -- both forms on a bucket-sized scratch buffer, three rounds each, the per-form times max-reduced over the ranks so that
    every rank takes the same decision (`mode_calibration` on the reducer, quoted by bench.py under config.comm).  No 8-GPU run has been
    available to the builder (the driver owns that node): the rule lets the first SCALE run pick the form the hardware prefers instead of
    the one an estimate prefers; rs_ag wins ties (within 3 %) because it is the form the sharded optimizer step builds on.
  * `VLP_DDP_MODE=sharded` (opt-in, round 4): reduce-scatter ONLY -- rank r keeps the mean of chunk r of every bucket -- and the
    optimizer step is sharded over the ranks (`ShardPlan`, used by FP16_Optimizer_State): each rank runs the grad-norm partial and the
    fused Adam update on its 1/W of master / m / v (1.39 GB of state and 0.62 ms of HBM-bound update per step shrink by W), the clip
    norm and the overflow flag are ONE 4-float all-reduce, and the updated fp16 PARAMETERS are all-gathered bucket by bucket (the same
    bytes as the gradient all-gather they replace) with per-bucket waits in the next forward.  The update arithmetic per element is the
    unsharded kernel's; parameters are bit-identical to the unsharded rs_ag run whenever the clip is inactive (the global norm is
    summed in another order).  Unmeasured on more than one GPU (world 2 over gloo on CPU and on one GPU, world 1 over RCCL).
"""
import os
import time

import torch
import torch.distributed as dist
from torch import nn


class _Stamp(object):
    """A point in time of a stream (HIP event) or, for CPU tensors (gloo tests), of the host."""

    def __init__(self, cuda):
        self.ev = torch.cuda.Event(enable_timing=True) if cuda else None
        self.t = None

    def record(self, stream=None):
        if self.ev is not None:
            self.ev.record(stream if stream is not None else torch.cuda.current_stream())
        else:
            self.t = time.perf_counter()
        return self

    def ms_until(self, other):
        if self.ev is not None:
            return self.ev.elapsed_time(other.ev)
        return (other.t - self.t) * 1e3


def coalesce_buckets(slices, cap):
    """Adjacent ready-slices [(lo, hi)] (in completion order) -> buckets of <= cap elements (a slice larger than cap stays whole).
    Returns (buckets, fire_at): bucket b is handed to the collective when slice index fire_at^-1(b) -- its LAST slice -- is ready."""
    buckets, fire_at = [], {}
    cur_lo, cur_hi = None, None
    for i, (lo, hi) in enumerate(slices):
        if cur_lo is None:
            cur_lo, cur_hi = lo, hi
        elif lo == cur_hi and (hi - cur_lo) <= cap:
            cur_hi = hi
        else:
            fire_at[i - 1] = len(buckets)
            buckets.append((cur_lo, cur_hi))
            cur_lo, cur_hi = lo, hi
    if cur_lo is not None:
        fire_at[len(slices) - 1] = len(buckets)
        buckets.append((cur_lo, cur_hi))
    return buckets, fire_at


def choose_mode(world, backend, divisible, calibrate=None, env=None):
    """The exchange-form rule (module docstring): -> (mode, why).  `calibrate()` -> {"allreduce": ms, "rs_ag": ms} (already agreed over the
    ranks) is only called when the rule needs a measurement.  Pure apart from that call: CPU-testable."""
    env = os.environ.get("VLP_DDP_MODE") if env is None else env
    if env:
        if env not in ("allreduce", "rs_ag", "sharded"):
            raise ValueError("VLP_DDP_MODE must be 'allreduce', 'rs_ag' or 'sharded', got %r" % env)
        return env, "VLP_DDP_MODE"
    force = os.environ.get("VLP_DDP_CALIBRATE") == "1"
    if not divisible:
        return "allreduce", "bucket sizes not divisible by the world size"
    if not force and (world < 4 or backend != "nccl"):
        return "allreduce", "fewer than 4 ranks or not RCCL: all-reduce"
    if calibrate is None:
        return "rs_ag", "4+ ranks on RCCL, no calibration available: the SURVEY 8e estimate"
    t = calibrate()
    mode = "rs_ag" if t["rs_ag"] <= 1.03 * t["allreduce"] else "allreduce"
    return mode, "measured at construction: all-reduce %.3f ms, reduce-scatter + all-gather %.3f ms per %.0f MB" % (t["allreduce"], t["rs_ag"], t.get("mb", 0.0))


class GradReducer(object):
    """Bucketed asynchronous all-reduce(mean) over slices of flat gradient buffers."""

    def __init__(self, flat_main, slices, flat_tail=None, process_group=None, bucket_cap_mb=50.0, mode=None):
        """flat_main: 1-D gradient buffer; slices: [(lo, hi)] in the order they become ready;
        flat_tail: a small buffer reduced at the end (biases / LayerNorm parameters);
        mode: "allreduce" (default) | "rs_ag" (env VLP_DDP_MODE)."""
        self.flat_main, self.flat_tail, self.pg = flat_main, flat_tail, process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        cap = int(bucket_cap_mb * 1024 * 1024 / flat_main.element_size())
        self.buckets, self.fire_at = coalesce_buckets(slices, cap)
        self._avg = dist.get_backend(process_group) == "nccl"
        self._work = []
        self.mode_calibration = None
        if mode is not None:
            if mode not in ("allreduce", "rs_ag", "sharded"):
                raise ValueError("mode must be 'allreduce', 'rs_ag' or 'sharded', got %r" % mode)
            self.mode, self.mode_why = mode, "constructor argument"
        else:
            sizes0 = [hi - lo for lo, hi in self.buckets] + ([flat_tail.numel()] if flat_tail is not None else [])
            self.mode, self.mode_why = choose_mode(self.world, dist.get_backend(process_group), all(n % self.world == 0 for n in sizes0), self._calibrate)
        if self.mode != "allreduce":
            # every bucket (and the tail) is cut into `world` equal chunks: checked HERE, not discovered per step (the engine lays
            # parameters out on 64-element boundaries, so any power-of-two world up to 64 divides)
            sizes = [hi - lo for lo, hi in self.buckets] + ([flat_tail.numel()] if flat_tail is not None else [])
            bad = [n for n in sizes if n % self.world]
            if bad:
                raise ValueError("VLP_DDP_MODE=%s: bucket sizes %r are not divisible by the world size %d" % (self.mode, bad, self.world))
        # comm profile (bench.py --gpus N, second un-timed pass): per collective the moment its slice was ready (stamp on the issuing
        # stream) and the moment it completed (stamp on an OBSERVER stream that waits for that collective only), plus the end of
        # backward's compute on the main stream -- so a scaling result can be read as exposed vs overlapped communication
        self.profile = False
        self._obs = None
        self._stamps = []          # [(label, numel, ready, done)] of the current step
        self.comm_steps = []       # one summary dict per profiled step (comm_summary())

    def _calibrate(self, rounds=3):
        """Both exchange forms on a scratch buffer of the largest bucket's size (values irrelevant), `rounds` timed rounds each after one
        warm-up; per-form time = the slowest rank's mean (one MAX all-reduce), so all ranks agree.  ~10-30 ms once per process."""
        n = max(hi - lo for lo, hi in self.buckets)
        n -= n % self.world
        buf = torch.zeros(n, device=self.flat_main.device, dtype=self.flat_main.dtype)
        cuda = buf.is_cuda
        saved, out = self.mode if hasattr(self, "mode") else None, {}
        for form in ("allreduce", "rs_ag"):
            self.mode = form
            ts = []
            for i in range(rounds + 1):
                if cuda:
                    torch.cuda.synchronize()
                dist.barrier(group=self.pg)
                t0 = time.perf_counter()
                self._reduce_impl(buf)
                for work, t in self._work:
                    if work is not None:
                        work.wait()
                self._work = []
                if cuda:
                    torch.cuda.synchronize()
                if i:
                    ts.append((time.perf_counter() - t0) * 1e3)
            out[form] = sum(ts) / len(ts)
        if saved is not None:
            self.mode = saved
        else:
            del self.mode
        tt = torch.tensor([out["allreduce"], out["rs_ag"]], device=buf.device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=self.pg)
        out = {"allreduce": float(tt[0]), "rs_ag": float(tt[1]), "mb": n * buf.element_size() / 2.0 ** 20}
        self.mode_calibration = {k: round(v, 4) for k, v in out.items()}
        return out

    def _observe(self, n_before, label, t, ready):
        """Profile hook: the collectives appended to self._work since n_before belong to one bucket; their completion is stamped on the
        observer stream (Work.wait() makes the CURRENT stream wait -- here the observer, which has nothing else to do)."""
        cuda = t.is_cuda
        done = _Stamp(cuda)
        if cuda:
            if self._obs is None:
                self._obs = torch.cuda.Stream(device=t.device)
            with torch.cuda.stream(self._obs):
                for work, _ in self._work[n_before:]:
                    if work is not None:
                        work.wait()
                done.record(self._obs)
        else:
            for work, _ in self._work[n_before:]:
                if work is not None:
                    work.wait()
            done.record()
        self._stamps.append((label, t.numel() * t.element_size(), ready, done))

    def _reduce(self, t, label="bucket"):
        if self.profile:
            n0, ready = len(self._work), _Stamp(t.is_cuda).record()
            self._reduce_impl(t)
            self._observe(n0, label, t, ready)
        else:
            self._reduce_impl(t)

    def _reduce_impl(self, t):
        if self.mode != "allreduce":                 # (also with one rank: the same RCCL calls, a path check)
            self._reduce_rs_ag(t, gather=self.mode == "rs_ag")
        elif self._avg:
            self._work.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.pg, async_op=True), None))
        else:
            self._work.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), t))

    def _reduce_rs_ag(self, t, gather=True):
        """Bucket = W equal chunks; rank r ends up owning the mean of chunk r (reduce-scatter, in place on its own chunk), then
        (gather=True) every rank gathers all chunks (all-gather, in place).  Both collectives are asynchronous on RCCL's stream; the
        all-gather is ordered after the reduce-scatter by that stream.  gather=False (sharded optimizer): the other chunks are left
        as they are -- nobody reads them, the owner of each chunk updates its parameters from it."""
        W, r = self.world, self.rank
        chunks = t.view(W, -1)
        if self._avg:
            self._work.append((dist.reduce_scatter_tensor(chunks[r], t, op=dist.ReduceOp.AVG, group=self.pg, async_op=True), None))
            if gather:
                self._work.append((dist.all_gather_into_tensor(t, chunks[r], group=self.pg, async_op=True), None))
        else:
            # gloo (CPU tests) has no reduce-scatter: W rooted reductions are the same data movement, then scale the owned chunk
            for dst in range(W):
                dist.reduce(chunks[dst], dst=dist.get_global_rank(self.pg, dst) if self.pg is not None else dst, op=dist.ReduceOp.SUM, group=self.pg)
            chunks[r].div_(W)
            if gather:
                dist.all_gather_into_tensor(t, chunks[r].clone(), group=self.pg)

    def bucket_ready(self, slice_index):
        b = self.fire_at.get(slice_index)
        if b is not None:
            lo, hi = self.buckets[b]
            self._reduce(self.flat_main[lo:hi], "bucket%d" % b)

    def finish(self):
        bwd_end = _Stamp(self.flat_main.is_cuda).record() if self.profile else None      # backward's last compute kernel, main stream
        if self.flat_tail is not None:
            self._reduce(self.flat_tail, "nodecay")
        for work, t in self._work:
            if work is not None:
                work.wait()          # makes the current stream wait for the collective; no host sync on RCCL
            if t is not None:
                t.div_(self.world)
        self._work = []
        if self.profile:
            resumed = _Stamp(self.flat_main.is_cuda).record()       # the main stream may continue (optimizer step) from here
            self._pending = (bwd_end, resumed, self._stamps)
            self._stamps = []

    def comm_collect(self):
        """Turn the stamps of the last profiled step into numbers (synchronises: call outside timed regions).  A collective's
        duration is counted from max(its slice was ready, the previous collective completed) -- RCCL runs them in issue order."""
        pend = getattr(self, "_pending", None)
        if pend is None:
            return None
        if self.flat_main.is_cuda:
            torch.cuda.synchronize()
        bwd_end, resumed, stamps = pend
        self._pending = None
        per, prev_done, total = [], None, 0.0
        for label, nbytes, ready, done in stamps:
            dur = ready.ms_until(done)
            if prev_done is not None:
                dur = min(dur, max(prev_done.ms_until(done), 0.0))
            per.append({"name": label, "mb": round(nbytes / 2.0 ** 20, 4), "ms": round(dur, 4), "done_after_backward_ms": round(bwd_end.ms_until(done), 4)})
            total += dur
            prev_done = done
        exposed = max(bwd_end.ms_until(resumed), 0.0)
        out = {"exposed_ms": round(exposed, 4), "overlapped_ms": round(max(total - exposed, 0.0), 4), "collectives_ms": round(total, 4), "per_bucket": per}
        self.comm_steps.append(out)
        return out

    def comm_summary(self):
        """Mean over the profiled steps: {"exposed_ms", "overlapped_ms", "collectives_ms", "per_bucket": [{name, mb, ms}], "steps"};
        exposed = end of backward's compute on the main stream -> the main stream may run the optimizer (all collectives waited for)."""
        if not self.comm_steps:
            return None
        n = len(self.comm_steps)
        out = {k: round(sum(s[k] for s in self.comm_steps) / n, 4) for k in ("exposed_ms", "overlapped_ms", "collectives_ms")}
        first = self.comm_steps[0]["per_bucket"]
        out["per_bucket"] = [{"name": b["name"], "mb": b["mb"], "ms": round(sum(s["per_bucket"][i]["ms"] for s in self.comm_steps) / n, 4),
                              "done_after_backward_ms": round(sum(s["per_bucket"][i]["done_after_backward_ms"] for s in self.comm_steps) / n, 4)}
                             for i, b in enumerate(first)]
        out["steps"] = n
        out["mode_rule"] = self.mode_why
        out["mode_calibration_ms"] = self.mode_calibration
        return out


def owned_chunk(lo, hi, world, rank):
    """Element range of chunk `rank` of the bucket [lo, hi) cut into `world` equal chunks."""
    n = (hi - lo) // world
    return lo + rank * n, lo + (rank + 1) * n


class ShardPlan(object):
    """Who owns what in the sharded optimizer step (VLP_DDP_MODE=sharded).  Pure bookkeeping over element ranges -- it runs on CPU
    tensors over gloo exactly as on the GPU over RCCL, so the partitioning, the norm exchange and the parameter all-gather are
    covered by the world-2 CPU tests with a torch stand-in for the update kernel."""

    def __init__(self, buckets, tail_numel, slices, world, rank, process_group=None):
        """buckets: coalesced [(lo, hi)] of the main (decay) buffer in completion order; tail_numel: size of the no-decay buffer;
        slices: the engine's ready-slices (to map `wait_params(slice index)` of the next forward onto a bucket)."""
        self.buckets, self.world, self.rank, self.pg = list(buckets), world, rank, process_group
        self.tail = (0, tail_numel)
        self.owned_main = [owned_chunk(lo, hi, world, rank) for lo, hi in self.buckets]
        self.owned_tail = owned_chunk(0, tail_numel, world, rank)
        self.bucket_of_slice = {}
        for i, (lo, hi) in enumerate(slices):
            self.bucket_of_slice[i] = [b for b, (blo, bhi) in enumerate(self.buckets) if blo <= lo and hi <= bhi][0]

    def owned(self, key):
        """[(lo, hi)] owned by this rank in the flat buffer `key` ("decay" | "nodecay")."""
        return list(self.owned_main) if key == "decay" else [self.owned_tail]

    def exchange_norms(self, stats):
        """stats: f32 [2 * groups] = (sum of squares, overflow flag) per param group over the OWNED ranges, in place -> the same over
        all ranks (flags add up: > 0 means some rank saw inf / nan).  One small all-reduce per step."""
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.pg)
        return stats

    def gather_params(self, flat_main, flat_tail):
        """All-gather the updated parameter chunks, bucket by bucket in the order the next forward reads them (no-decay buffer first,
        then the LAST bucket -- embeddings / region projections -- down to bucket 0, the task head).  Returns {"nodecay" | bucket:
        work-or-None}; the engine waits per bucket (Engine.wait_params)."""
        works = {}
        nccl = dist.get_backend(self.pg) == "nccl"

        def gather(t):
            chunks = t.view(self.world, -1)
            if nccl:
                return dist.all_gather_into_tensor(t, chunks[self.rank], group=self.pg, async_op=True)
            dist.all_gather_into_tensor(t, chunks[self.rank].clone(), group=self.pg)      # gloo: no in-place aliasing
            return None
        works["nodecay"] = gather(flat_tail)
        for b in reversed(range(len(self.buckets))):
            lo, hi = self.buckets[b]
            works[b] = gather(flat_main[lo:hi])
        return works

    def gather_state(self, tensors_main, tensors_tail):
        """Checkpointing: every rank's fp32 master / m / v are only current on its own chunks -- gather them (blocking; rare)."""
        for t in tensors_main:
            for lo, hi in self.buckets:
                sl = t[lo:hi]
                dist.all_gather_into_tensor(sl, sl.view(self.world, -1)[self.rank].clone(), group=self.pg)
        for t in tensors_tail:
            dist.all_gather_into_tensor(t, t.view(self.world, -1)[self.rank].clone(), group=self.pg)


class DistributedDataParallel(nn.Module):
    """Same constructor surface as torch's DDP for the arguments the reference passes; exposes `.module`."""

    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=False, process_group=None,
                 bucket_cap_mb=50.0, **unused_kwargs):
        super(DistributedDataParallel, self).__init__()
        if not dist.is_initialized():
            raise RuntimeError("init_process_group() must be called before wrapping the model (run_img2txt_dist.py:233)")
        self.module = module
        eng = module.engine
        eng.pack()
        self.engine = eng
        for key in ("decay", "nodecay"):                     # initial parameter sync (DDP broadcasts rank 0's state)
            dist.broadcast(eng.flat[key], src=0, group=process_group)
        self.reducer = GradReducer(eng.gflat["decay"], eng.buckets, eng.gflat["nodecay"], process_group, bucket_cap_mb)
        eng.grad_ready_hook = self.reducer.bucket_ready
        eng.post_backward_hook = self.reducer.finish
        # sharded optimizer step: the optimizer (FP16_Optimizer_State) finds the plan on the engine
        eng.shard_plan = None
        if self.reducer.mode == "sharded":
            eng.shard_plan = ShardPlan(self.reducer.buckets, eng.gflat["nodecay"].numel(), eng.buckets, self.reducer.world, self.reducer.rank,
                                       process_group)

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)
