"""CPU, world_size 2 over gloo: the bucketed gradient reducer used by vlp_amd.distributed.DistributedDataParallel
(the N > 1 path of bench.py / run_img2txt_dist.py).  The engine is emulated by firing the ready-hooks in
completion order on flat CPU buffers."""
import os
import tempfile

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vlp_amd.distributed import GradReducer


def _worker(rank, world, init_file, q):
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        n_main, n_tail = 5000, 300
        slices = [(0, 700), (700, 2100), (2100, 3500), (3500, 5000)]
        for dtype, cap_mb in ((torch.float32, 0.008), (torch.float16, 100.0)):
            main = torch.randn(n_main).to(dtype)
            tail = torch.randn(n_tail).to(dtype)
            ref_main, ref_tail = main.clone().float(), tail.clone().float()
            dist.all_reduce(ref_main)
            dist.all_reduce(ref_tail)
            red = GradReducer(main, slices, tail, None, bucket_cap_mb=cap_mb)
            # small cap -> several buckets; big cap -> one bucket covering everything
            assert red.buckets[0][0] == 0 and red.buckets[-1][1] == n_main
            assert sum(hi - lo for lo, hi in red.buckets) == n_main
            for i in range(len(slices)):
                red.bucket_ready(i)
            red.finish()
            tol = 1e-6 if dtype == torch.float32 else 2e-3
            assert torch.allclose(main.float(), ref_main / world, atol=tol, rtol=tol)
            assert torch.allclose(tail.float(), ref_tail / world, atol=tol, rtol=tol)
            # gradient accumulation: reducing an already-averaged + new local gradient keeps the averaged part
            acc = main.clone()
            local = torch.randn(n_main).to(dtype)
            buf = (acc.float() + local.float()).to(dtype)
            ref = local.clone().float()
            dist.all_reduce(ref)
            red2 = GradReducer(buf, slices, None, None, bucket_cap_mb=cap_mb)
            for i in range(len(slices)):
                red2.bucket_ready(i)
            red2.finish()
            assert torch.allclose(buf.float(), acc.float() + ref / world, atol=10 * tol, rtol=10 * tol)
            # communication profile (bench.py --gpus N fills config.comm from it): the fields exist, one entry per collective in issue
            # order, and the parts add up; results of a profiled reduction are the unprofiled ones
            main2, tail2 = torch.randn(n_main).to(dtype), torch.randn(n_tail).to(dtype)
            want_main = main2.clone().float()
            dist.all_reduce(want_main)
            red3 = GradReducer(main2, slices, tail2, None, bucket_cap_mb=cap_mb)
            red3.profile = True
            for step in range(2):
                for i in range(len(slices)):
                    red3.bucket_ready(i)
                red3.finish()
                one = red3.comm_collect()
                assert one is not None and len(one["per_bucket"]) == len(red3.buckets) + 1
                if step == 0:
                    assert torch.allclose(main2.float(), want_main / world, atol=tol, rtol=tol)
            summ = red3.comm_summary()
            assert summ["steps"] == 2 and [b["name"] for b in summ["per_bucket"]] == ["bucket%d" % i for i in range(len(red3.buckets))] + ["nodecay"]
            assert summ["exposed_ms"] >= 0 and summ["overlapped_ms"] >= 0 and summ["collectives_ms"] >= 0
            assert abs(summ["collectives_ms"] - sum(b["ms"] for b in summ["per_bucket"])) <= 1e-2 + 1e-3 * summ["collectives_ms"]
            assert all(b["mb"] > 0 for b in summ["per_bucket"])
        # the exchange-form rule, measured branch (forced: gloo / world 2 would take all-reduce unmeasured): both forms are timed on a scratch
        # buffer, the ranks agree on ONE mode, and the reducer built on it reduces correctly
        os.environ["VLP_DDP_CALIBRATE"] = "1"
        try:
            main4 = torch.randn(4096)
            want4 = main4.clone()
            dist.all_reduce(want4)
            red4 = GradReducer(main4, [(0, 2048), (2048, 4096)], torch.zeros(8), None, bucket_cap_mb=0.008)
        finally:
            del os.environ["VLP_DDP_CALIBRATE"]
        assert red4.mode in ("allreduce", "rs_ag") and red4.mode_calibration is not None and "measured at construction" in red4.mode_why
        modes = [None, None]
        dist.all_gather_object(modes, red4.mode)
        assert modes[0] == modes[1]
        red4.bucket_ready(0)
        red4.bucket_ready(1)
        red4.finish()
        assert torch.allclose(main4, want4 / world, atol=1e-6)
        q.put((rank, "ok"))
    except Exception as e:   # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_exchange_form_rule():
    """VERDICT r5 #7b: the documented rule, without a process group.  Environment override first; all-reduce below four ranks / off RCCL /
    when a bucket does not divide; from four RCCL ranks the measured times decide, reduce-scatter + all-gather winning ties (3 %)."""
    from vlp_amd.distributed import choose_mode
    never = lambda: (_ for _ in ()).throw(AssertionError("no measurement expected"))      # noqa: E731
    assert choose_mode(8, "nccl", True, never, env="sharded")[0] == "sharded"
    assert choose_mode(2, "nccl", True, never, env="")[0] == "allreduce"
    assert choose_mode(8, "gloo", True, never, env="")[0] == "allreduce"
    assert choose_mode(8, "nccl", False, never, env="")[0] == "allreduce"
    assert choose_mode(8, "nccl", True, lambda: {"allreduce": 1.00, "rs_ag": 0.40, "mb": 50.0}, env="")[0] == "rs_ag"
    assert choose_mode(4, "nccl", True, lambda: {"allreduce": 1.00, "rs_ag": 1.02, "mb": 50.0}, env="")[0] == "rs_ag"
    mode, why = choose_mode(8, "nccl", True, lambda: {"allreduce": 0.50, "rs_ag": 0.80, "mb": 50.0}, env="")
    assert mode == "allreduce" and "0.500" in why and "0.800" in why
    assert choose_mode(8, "nccl", True, None, env="")[0] == "rs_ag"
    with pytest.raises(ValueError):
        choose_mode(8, "nccl", True, never, env="ring")


def test_grad_reducer_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "nonexistent_file")     # same file:// rendezvous style as the reference (:162)
        procs = [ctx.Process(target=_worker, args=(r, world, init_file, q)) for r in range(world)]
        for p in procs:
            p.start()
        results = [q.get(timeout=120) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


# ---------------------------------------------------------------------------------------------------------------------
# the REAL 12-layer bucket plan (no GPU needed: Engine.plan_layout is a pure function of the parameter shapes)
# ---------------------------------------------------------------------------------------------------------------------
def test_real_12_layer_bucket_plan():
    from vlp_amd.distributed import coalesce_buckets
    from vlp_amd.engine import ALIGN, is_no_decay
    from vlp_amd.modeling import BertConfig, BertForPreTrainingLossMask
    for tasks in ("img2txt", "vqa2"):
        cfg = BertConfig(28996, num_hidden_layers=12, type_vocab_size=6)
        model = BertForPreTrainingLossMask(cfg, enable_butd=True, len_vis_input=100, tasks=tasks, allow_random_fc7=True)
        lay = model.engine.plan_layout(model)
        names, offs, sizes, slices = lay["names"], lay["offsets"], lay["sizes"], lay["buckets"]
        numel = {n: p.numel() for n, p in model.named_parameters()}
        # every parameter exactly once, decay / no-decay split as the train script's two groups (run_img2txt_dist.py:394-401)
        assert sorted(names["decay"] + names["nodecay"]) == sorted(numel)
        assert all(not is_no_decay(n) for n in names["decay"]) and all(is_no_decay(n) for n in names["nodecay"])
        assert len(numel) == (214 if tasks == "vqa2" else 210)
        # parameters do not overlap, start on 128-byte boundaries, and lie in list order
        for grp in ("decay", "nodecay"):
            end = 0
            for n in names[grp]:
                assert offs[grp][n] % ALIGN == 0 and offs[grp][n] >= end, n
                end = offs[grp][n] + numel[n]
            assert end <= sizes[grp]
        # ready-slices: contiguous, in backward-completion order, covering the decay buffer exactly once
        assert len(slices) == 12 + 3
        assert slices[0][0] == 0 and slices[-1][1] == sizes["decay"]
        for (lo, hi), (lo2, _) in zip(slices, slices[1:]):
            assert lo < hi and hi == lo2
        # slice 0 = task head, slices 1..12 = layers 11..0 (six weight matrices each), 13 = embedding tables (final after embed_bwd),
        # 14 = region projections (final after the last wgrads of backward)
        def owner(n):
            o = offs["decay"][n]
            return [i for i, (lo, hi) in enumerate(slices) if lo <= o < hi][0]
        for i in range(12):
            L = "bert.encoder.layer.%d." % i
            for suffix in ("output.dense.weight", "intermediate.dense.weight", "attention.output.dense.weight", "attention.self.query.weight",
                           "attention.self.key.weight", "attention.self.value.weight"):
                assert owner(L + suffix) == 12 - i, (L + suffix, owner(L + suffix))
            q, k, v = (offs["decay"][L + "attention.self.%s.weight" % x] for x in ("query", "key", "value"))
            assert k == q + 768 * 768 and v == k + 768 * 768           # packed QKV GEMM reads them as one [2304, 768] matrix
        assert owner("bert.embeddings.word_embeddings.weight") == 13 and owner("vis_embed.0.weight") == 14
        # the 45 MB embedding slice is its own bucket: its reduction starts before the region-projection wgrads, not after them
        bk, fa = coalesce_buckets(slices, int(50.0 * 1024 * 1024 / 2))
        assert bk[fa[13]] == slices[13], (bk[fa[13]], slices[13])
        assert owner("ans_classifier.0.weight" if tasks == "vqa2" else "cls.predictions.transform.dense.weight") == 0
        # coalescing to <= 50 MB of fp16: every bucket within the cap unless it is a single slice; buckets tile the buffer; a bucket
        # fires when its last slice is ready, in order
        cap = int(50.0 * 1024 * 1024 / 2)
        buckets, fire_at = coalesce_buckets(slices, cap)
        assert buckets[0][0] == 0 and buckets[-1][1] == sizes["decay"]
        for (lo, hi), (lo2, _) in zip(buckets, buckets[1:]):
            assert hi == lo2
        single = {(lo, hi) for lo, hi in slices}
        for b in buckets:
            assert (b[1] - b[0]) <= cap or b in single, b
        assert sorted(fire_at.values()) == list(range(len(buckets)))
        fired = [b for _, b in sorted(fire_at.items())]
        assert fired == sorted(fired)
        for si, b in fire_at.items():
            assert slices[si][1] == buckets[b][1]
        assert 5 <= len(buckets) <= 8, len(buckets)        # 231.9 MB of fp16 gradients in <= 50 MB pieces (the 45 MB embedding slice and the region projections are separate buckets)
        # rs_ag mode needs bucket sizes divisible by the world size (8)
        assert all((hi - lo) % 8 == 0 for lo, hi in buckets) and sizes["nodecay"] % 8 == 0


# ---------------------------------------------------------------------------------------------------------------------
# DistributedDataParallel's hook wiring, world 2 over gloo, through a fake engine (flat CPU buffers, hooks fired in order)
# ---------------------------------------------------------------------------------------------------------------------
class _FakeEngine(object):
    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        self.flat = {"decay": torch.randn(4096, generator=g), "nodecay": torch.randn(256, generator=g)}
        self.gflat = {k: torch.zeros_like(v) for k, v in self.flat.items()}
        self.buckets = [(0, 1024), (1024, 2048), (2048, 3072), (3072, 4096)]
        self.grad_ready_hook = None
        self.post_backward_hook = None
        self.packed = False

    def pack(self):
        self.packed = True

    def backward(self, gd, gn):
        """what Engine.backward does with the hooks: write a slice, announce it; finish after the last."""
        for i, (lo, hi) in enumerate(self.buckets):
            self.gflat["decay"][lo:hi] = gd[lo:hi]
            self.grad_ready_hook(i)
        self.gflat["nodecay"].copy_(gn)
        self.post_backward_hook()


class _FakeModule(torch.nn.Module):
    def __init__(self, seed):
        super(_FakeModule, self).__init__()
        self.engine = _FakeEngine(seed)

    def forward(self, x):
        return x + 1


def _ddp_worker(rank, world, init_file, mode, q):
    os.environ["VLP_DDP_MODE"] = mode
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    try:
        from vlp_amd.distributed import DistributedDataParallel as DDP
        mod = _FakeModule(seed=10 + rank)               # ranks start from DIFFERENT parameters ...
        p0 = [t.clone() for t in (mod.engine.flat["decay"], mod.engine.flat["nodecay"])]
        ddp = DDP(mod, device_ids=None, find_unused_parameters=True, bucket_cap_mb=0.01)
        eng = mod.engine
        assert eng.packed and ddp.module is mod and ddp.reducer.mode == mode
        assert float(ddp(torch.zeros(1))) == 1.0
        # ... and hold rank 0's after construction (DDP's initial broadcast)
        ref = [t.clone() for t in p0]
        for t in ref:
            dist.broadcast(t, src=0)
        assert torch.equal(eng.flat["decay"], ref[0]) and torch.equal(eng.flat["nodecay"], ref[1])
        assert eng.grad_ready_hook is not None and eng.post_backward_hook is not None
        assert len(ddp.reducer.buckets) == 2               # 4 slices of 1024 fp32 coalesced under a 0.01 MB cap: 2 x 2048
        for step in range(3):
            g = torch.Generator().manual_seed(1000 * step + rank)
            gd, gn = torch.randn(4096, generator=g), torch.randn(256, generator=g)
            if step == 2 and rank == 1:
                gd[77] = float("inf")                      # an fp16 overflow on ONE rank ...
            want_d, want_n = gd.clone(), gn.clone()
            dist.all_reduce(want_d)
            dist.all_reduce(want_n)
            eng.backward(gd, gn)
            if mode == "sharded":
                # reduce-scatter only: this rank's chunk of every bucket (and of the tail) holds the mean, and the plan the optimizer
                # will use says exactly which elements those are
                plan = eng.shard_plan
                assert plan is not None and plan.world == world and plan.rank == rank
                if step < 2:
                    for lo, hi in plan.owned("decay"):
                        assert torch.allclose(eng.gflat["decay"][lo:hi], want_d[lo:hi] / world, atol=1e-6)
                    lo, hi = plan.owned("nodecay")[0]
                    assert torch.allclose(eng.gflat["nodecay"][lo:hi], want_n[lo:hi] / world, atol=1e-6)
                    assert sum(hi - lo for lo, hi in plan.owned("decay")) * world == 4096
                continue
            assert getattr(eng, "shard_plan", None) is None
            if step < 2:
                assert torch.allclose(eng.gflat["decay"], want_d / world, atol=1e-6)
                assert torch.allclose(eng.gflat["nodecay"], want_n / world, atol=1e-6)
            else:
                # ... is non-finite in the SAME slot on every rank after the reduction, so every rank's overflow check
                # (vlp_sumsq's inf/nan flag on the reduced buffer) takes the same skip decision: no extra flag exchange
                bad = ~torch.isfinite(eng.gflat["decay"])
                assert bool(bad[77]) and int(bad.sum()) == 1
                flag = torch.tensor([float(bad.any())])
                both = [torch.zeros(1) for _ in range(world)]
                dist.all_gather(both, flag)
                assert all(float(b) == 1.0 for b in both)
        q.put((rank, "ok"))
    except Exception as e:   # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run_world2(target, *extra):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "nonexistent_file")
        procs = [ctx.Process(target=target, args=(r, world, init_file) + extra + (q,)) for r in range(world)]
        for p in procs:
            p.start()
        results = [q.get(timeout=180) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_ddp_hook_wiring_world2_gloo_allreduce():
    _run_world2(_ddp_worker, "allreduce")


def test_ddp_hook_wiring_world2_gloo_reduce_scatter_all_gather():
    _run_world2(_ddp_worker, "rs_ag")


# ---------------------------------------------------------------------------------------------------------------------
# sharded optimizer step (VLP_DDP_MODE=sharded): partitioning, norm exchange, parameter all-gather -- with a torch stand-in
# for the fused Adam kernel (per element the product runs the same kernel on an element range; it has no CPU form)
# ---------------------------------------------------------------------------------------------------------------------
def _adam_range(p32, m, v, g16, p16, lo, hi, combined, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, decay=0.01):
    g = g16[lo:hi].float() / combined
    m[lo:hi].mul_(b1).add_(g, alpha=1 - b1)
    v[lo:hi].mul_(b2).add_(g * g, alpha=1 - b2)
    p32[lo:hi].sub_(lr * (m[lo:hi] / (torch.sqrt(v[lo:hi]) + eps) + decay * p32[lo:hi]))
    p16[lo:hi].copy_(p32[lo:hi].half())


def _sharded_worker(rank, world, init_file, q):
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    try:
        from vlp_amd.distributed import ShardPlan, owned_chunk
        n_main, n_tail = 8192, 512
        slices = [(0, 1024), (1024, 3072), (3072, 6144), (6144, 8192)]
        scale = 1024.0
        for clip_active in (False, True):
            g = torch.Generator().manual_seed(7)
            p16 = [(torch.randn(n, generator=g) * 0.05).half() for n in (n_main, n_tail)]          # identical parameters on every rank
            gl = torch.Generator().manual_seed(100 + rank)
            amp = (30.0 if clip_active else 0.002) * scale
            grads = [(torch.randn(n, generator=gl) * amp / (n ** 0.5)).half() for n in (n_main, n_tail)]
            runs = {}
            for mode in ("rs_ag", "sharded"):
                gm, gt = grads[0].clone(), grads[1].clone()
                pm, pt = p16[0].clone(), p16[1].clone()
                state = [[t.float(), torch.zeros(t.numel()), torch.zeros(t.numel())] for t in (pm, pt)]      # master, m, v
                red = GradReducer(gm, slices, gt, None, bucket_cap_mb=0.008, mode=mode)
                assert len(red.buckets) >= 2
                for i in range(len(slices)):
                    red.bucket_ready(i)
                red.finish()
                if mode == "rs_ag":                    # replicated step: every rank, whole buffers
                    stats = torch.tensor([float((gm.float() ** 2).sum()), 0.0, float((gt.float() ** 2).sum()), 0.0])
                    ranges = {"decay": [(0, n_main)], "nodecay": [(0, n_tail)]}
                else:
                    plan = ShardPlan(red.buckets, n_tail, slices, world, rank)
                    # ownership: the chunks tile every bucket exactly once over the ranks
                    for (lo, hi), (olo, ohi) in zip(red.buckets, plan.owned_main):
                        assert (olo, ohi) == owned_chunk(lo, hi, world, rank) and (ohi - olo) * world == hi - lo
                    assert all(plan.buckets[plan.bucket_of_slice[i]][0] <= lo and hi <= plan.buckets[plan.bucket_of_slice[i]][1]
                               for i, (lo, hi) in enumerate(slices))
                    ranges = {"decay": plan.owned("decay"), "nodecay": plan.owned("nodecay")}
                    stats = torch.tensor([sum(float((gm[lo:hi].float() ** 2).sum()) for lo, hi in ranges["decay"]), 0.0,
                                          sum(float((gt[lo:hi].float() ** 2).sum()) for lo, hi in ranges["nodecay"]), 0.0])
                    plan.exchange_norms(stats)
                for gi, (key, gbuf, pbuf) in enumerate((("decay", gm, pm), ("nodecay", gt, pt))):
                    norm = float(stats[2 * gi]) ** 0.5
                    clip = (norm / scale + 1e-6) / 1.0
                    assert (clip > 1) == clip_active, (clip, clip_active)
                    combined = clip * scale if clip > 1 else scale
                    for lo, hi in ranges[key]:
                        _adam_range(state[gi][0], state[gi][1], state[gi][2], gbuf, pbuf, lo, hi, combined)
                if mode == "sharded":
                    works = plan.gather_params(pm, pt)
                    assert set(works) == {"nodecay"} | set(range(len(red.buckets)))
                    plan.gather_state([state[0][0], state[0][1], state[0][2]], [state[1][0], state[1][1], state[1][2]])
                runs[mode] = (pm, pt, state)
            for a, b in zip(runs["rs_ag"][:2], runs["sharded"][:2]):
                if clip_active:        # the global norm is summed in another order: the clip factor may move by an ulp
                    assert torch.allclose(a.float(), b.float(), rtol=2e-3, atol=1e-6)
                else:                  # clip inactive: bit-identical parameters
                    assert torch.equal(a, b)
            if not clip_active:
                for sa, sb in zip(runs["rs_ag"][2], runs["sharded"][2]):
                    for ta, tb in zip(sa, sb):
                        assert torch.equal(ta, tb)      # gathered master / m / v = the replicated state
            # every rank holds the same parameters after the sharded step
            both = [torch.zeros_like(runs["sharded"][0]) for _ in range(world)]
            dist.all_gather(both, runs["sharded"][0])
            assert torch.equal(both[0], both[1])
        # divisibility is checked at construction, not discovered per step
        try:
            GradReducer(torch.zeros(6146), [(0, 3073), (3073, 6146)], torch.zeros(7), None, bucket_cap_mb=0.001, mode="sharded")
            raise AssertionError("expected ValueError")
        except ValueError:
            pass
        q.put((rank, "ok"))
    except Exception as e:   # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_sharded_optimizer_step_world2_gloo():
    _run_world2(_sharded_worker)


def test_ddp_hook_wiring_world2_gloo_sharded():
    _run_world2(_ddp_worker, "sharded")


def test_bench_self_spawn_for_n_gpus(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (one rank per GPU, 127.0.0.1
    rendezvous, the caller's arguments preserved); on a node with fewer GPUs it fails at the device count with a clear message."""
    import importlib
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3"], capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "exposes" in (r.stdout + r.stderr) and "one rank per GPU" in (r.stdout + r.stderr), (r.stdout, r.stderr)
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    with pytest.raises(SystemExit) as e:
        bench.self_spawn(4, ["--gpus", "4", "--steps", "3"])
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
