"""N3 measurement: can the packed-store prefetcher feed the training step?  Builds a synthetic packed store under /tmp, then times
(a) the prefetcher alone, (b) the 12-layer training step on device-resident synthetic batches, (c) the same step fed by the prefetcher."""
import json
import os
import random
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vlp_amd import synthetic as S  # noqa: E402
from vlp_amd.data import BatchPrefetcher, PackedRegionStore, TextPreprocessor, write_packed  # noqa: E402
from vlp_amd.modeling import BertConfig, BertForPreTrainingLossMask  # noqa: E402
from vlp_amd.optimization_fp16 import FP16_Optimizer_State, FusedAdam  # noqa: E402
from vlp_amd.run_img2txt_dist import train_step  # noqa: E402

dev = torch.device("cuda:0")
N, B, STEPS = int(os.environ.get("N_IMAGES", 1024)), 64, int(os.environ.get("STEPS", 40))
rng = np.random.RandomState(0)
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    feats = np.abs(rng.standard_normal((N, 100, 2048))).astype(np.float16)
    cls = rng.rand(N, 100, 1601).astype(np.float16)
    xy1 = rng.uniform(0, 400, size=(N, 100, 2))
    box = np.concatenate((xy1, xy1 + rng.uniform(10, 200, size=(N, 100, 2)), rng.rand(N, 100, 1), rng.uniform(0.2, 1, size=(N, 100, 1))), axis=2).astype(np.float32)
    ids = ["img%06d" % i for i in range(N)]
    write_packed(d, ids, feats, cls, box)
    del feats, cls, box
    store = PackedRegionStore(d)
    examples = [(ids[i % N], rng.randint(1000, 28000, size=rng.randint(6, 64)).tolist()) for i in range(5 * N)]
    kw = dict(max_pred=3, mask_prob=0.15, vocab_size=28996, cls_id=S.CLS_ID, sep_id=S.SEP_ID, mask_id=S.MASK_ID, unk_id=S.UNK_ID, max_len=167, max_len_b=64)
    p_s2s, p_bi = TextPreprocessor(mode="s2s", **kw), TextPreprocessor(mode="bi", **kw)
    out = {"batch": B, "images": N, "bytes_per_sample_h2d": 100 * 2048 * 2 + 100 * 1601 * 2 + 100 * 6 * 4 + 2 * 167 * 8 + 9 * 8 + 12 + 8}
    # (a) loader alone, over worker-thread counts (WORKERS="1,2,4,8"); host cores of this process stated beside it
    out["host_cpus"] = len(os.sched_getaffinity(0))
    out["loader_only_samples_per_s"] = {}
    for w in [int(x) for x in os.environ.get("WORKERS", "1,2,4,8").split(",")]:
        pf = BatchPrefetcher(store, examples, B, p_s2s, p_bi, s2s_prob=0.75, device=dev, steps=STEPS, seed=0, num_workers=w)
        for _ in pf:                # warm the page cache / pinned slots once
            break
        t0 = time.perf_counter()
        n = 0
        for batch in pf:
            n += 1
        torch.cuda.synchronize()
        out["loader_only_samples_per_s"][str(w)] = round(n * B / (time.perf_counter() - t0), 1)
    NW = int(os.environ.get("TRAIN_WORKERS", 4))
    out["train_workers"] = NW
    # model
    cfg = BertConfig(28996, num_hidden_layers=12, type_vocab_size=6)
    model = BertForPreTrainingLossMask(cfg, enable_butd=True, len_vis_input=100, tasks="img2txt", allow_random_fc7=True).half().to(dev).train()
    named = list(model.named_parameters())
    nd = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    groups = [{"params": [p for n_, p in named if not any(x in n_ for x in nd)], "weight_decay": 0.01},
              {"params": [p for n_, p in named if any(x in n_ for x in nd)], "weight_decay": 0.0}]
    opt = FP16_Optimizer_State(FusedAdam(groups, lr=1e-4, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True)
    resident = S.batch_to(S.make_batch(B, max_len_b=64, vocab_size=28996, max_pred=3, seed=1), dev, half=True)
    for _ in range(5):
        train_step(model, opt, resident, 1e-4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        train_step(model, opt, resident, 1e-4)
    torch.cuda.synchronize()
    out["train_resident_samples_per_s"] = round(STEPS * B / (time.perf_counter() - t0), 1)
    # (c) fed by the prefetcher (first batches warm the MaskSpec / RawRegions paths)
    it = iter(BatchPrefetcher(store, examples, B, p_s2s, p_bi, s2s_prob=0.75, device=dev, steps=STEPS + 5, seed=1))
    for _ in range(5):
        train_step(model, opt, next(it), 1e-4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    for batch in it:
        train_step(model, opt, batch, 1e-4)
        k += 1
    torch.cuda.synchronize()
    out["train_from_prefetcher_samples_per_s"] = round(k * B / (time.perf_counter() - t0), 1)
    print(json.dumps(out))
