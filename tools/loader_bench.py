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
    eng = model.engine
    resident = S.batch_to(S.make_batch(B, max_len_b=64, vocab_size=28996, max_pred=3, seed=1), dev, half=True)

    def timed(batches, n):
        t0 = time.perf_counter()
        k = 0
        for b in batches:
            train_step(model, opt, b, 1e-4)
            k += 1
            if k == n:
                break
        torch.cuda.synchronize()
        return round(k * B / (time.perf_counter() - t0), 1)

    def repeat(b):
        while True:
            yield b

    # (b) resident batches: dense (the reference's [B, L, L] mask tensor, all L positions) and padding-free (a prefetcher batch kept resident:
    #     MaskSpec carries the lengths, so the step packs)
    eng.varlen = False
    timed(repeat(resident), 5)
    out["train_resident_dense_samples_per_s"] = timed(repeat(resident), STEPS)
    eng.varlen = "auto"
    # the SAME batches the prefetcher arm below will deliver (seed 1, its steps 5 .. 5 + STEPS), kept resident: equal rows, equal length tuples
    keep = []
    for i, b in enumerate(BatchPrefetcher(store, examples, B, p_s2s, p_bi, s2s_prob=0.75, device=dev, steps=STEPS + 5, seed=1, num_workers=NW)):
        if i >= 5:
            keep.append(tuple(t.clone() if torch.is_tensor(t) else type(t)(*(x.clone() if torch.is_tensor(x) else x for x in t)) for t in b))
    timed(iter(keep), 5)
    eng._pk_cache.clear()                 # every timed step builds its row map, as a real epoch does
    out["train_resident_padding_free_samples_per_s"] = timed(iter(keep), STEPS)
    out["padding_free_rows_mean"] = round(sum(sum(b[2].lens_host) for b in keep) / len(keep), 1)
    del keep
    # (c) fed by the prefetcher (NW loader threads), padding-free (the default for MaskSpec batches) and dense
    for name, mode in (("padding_free", "auto"), ("dense", False)):
        eng.varlen = mode
        it = iter(BatchPrefetcher(store, examples, B, p_s2s, p_bi, s2s_prob=0.75, device=dev, steps=STEPS + 5, seed=1, num_workers=NW))
        timed(it, 5)
        out["train_from_prefetcher_%s_samples_per_s" % name] = timed(it, STEPS)
        for _ in it:
            pass
    print(json.dumps(out))
