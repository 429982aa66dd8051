"""CPU, world_size 2 over gloo: the bucketed gradient reducer used by vlp_amd.distributed.DistributedDataParallel
(the N > 1 path of bench.py / run_img2txt_dist.py).  The engine is emulated by firing the ready-hooks in
completion order on flat CPU buffers."""
import os
import tempfile

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
        q.put((rank, "ok"))
    except Exception as e:   # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


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
