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
  * `VLP_DDP_MODE=rs_ag` (opt-in) issues every bucket as reduce-scatter + all-gather instead of one all-reduce (SURVEY.md
    section 5 / 8e: on point-to-point xGMI a direct reduce-scatter/all-gather uses all 7 links of a GPU, a ring is bound by one);
    same result up to fp16 summation order.  No scaling curve has been measured yet (the driver owns the 8-GPU runs), so
    all-reduce stays the default.
"""
import os

import torch.distributed as dist
from torch import nn


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


class GradReducer(object):
    """Bucketed asynchronous all-reduce(mean) over slices of flat gradient buffers."""

    def __init__(self, flat_main, slices, flat_tail=None, process_group=None, bucket_cap_mb=50.0, mode=None):
        """flat_main: 1-D gradient buffer; slices: [(lo, hi)] in the order they become ready;
        flat_tail: a small buffer reduced at the end (biases / LayerNorm parameters);
        mode: "allreduce" (default) | "rs_ag" (env VLP_DDP_MODE)."""
        self.flat_main, self.flat_tail, self.pg = flat_main, flat_tail, process_group
        self.mode = mode or os.environ.get("VLP_DDP_MODE", "allreduce")
        if self.mode not in ("allreduce", "rs_ag"):
            raise ValueError("VLP_DDP_MODE must be 'allreduce' or 'rs_ag', got %r" % self.mode)
        self.world = dist.get_world_size(process_group)
        cap = int(bucket_cap_mb * 1024 * 1024 / flat_main.element_size())
        self.buckets, self.fire_at = coalesce_buckets(slices, cap)
        self._avg = dist.get_backend(process_group) == "nccl"
        self._work = []

    def _reduce(self, t):
        if self.mode == "rs_ag" and t.numel() % self.world == 0:      # (also with one rank: the same RCCL calls, a path check)
            self._reduce_rs_ag(t)
        elif self._avg:
            self._work.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.pg, async_op=True), None))
        else:
            self._work.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), t))

    def _reduce_rs_ag(self, t):
        """Bucket = W equal chunks; rank r ends up owning the mean of chunk r (reduce-scatter, in place on its own chunk), then
        every rank gathers all chunks (all-gather, in place).  Both collectives are asynchronous on RCCL's stream; the all-gather
        is ordered after the reduce-scatter by that stream."""
        W, r = self.world, dist.get_rank(self.pg)
        chunks = t.view(W, -1)
        if self._avg:
            self._work.append((dist.reduce_scatter_tensor(chunks[r], t, op=dist.ReduceOp.AVG, group=self.pg, async_op=True), None))
            self._work.append((dist.all_gather_into_tensor(t, chunks[r], group=self.pg, async_op=True), None))
        else:
            # gloo (CPU tests) has no reduce-scatter: W rooted reductions are the same data movement, then scale the owned chunk
            for dst in range(W):
                dist.reduce(chunks[dst], dst=dist.get_global_rank(self.pg, dst) if self.pg is not None else dst, op=dist.ReduceOp.SUM, group=self.pg)
            chunks[r].div_(W)
            dist.all_gather_into_tensor(t, chunks[r].clone(), group=self.pg)

    def bucket_ready(self, slice_index):
        b = self.fire_at.get(slice_index)
        if b is not None:
            lo, hi = self.buckets[b]
            self._reduce(self.flat_main[lo:hi])

    def finish(self):
        if self.flat_tail is not None:
            self._reduce(self.flat_tail)
        for work, t in self._work:
            work.wait()          # makes the current stream wait for the collective; no host sync on RCCL
            if t is not None:
                t.div_(self.world)
        self._work = []


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

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)
