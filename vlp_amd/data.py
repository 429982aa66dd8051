"""Region-feature store and batch pipeline for the HIP engine.

The reference opens three HDF5 files PER SAMPLE inside `Preprocess4Seq2seq.__call__` (vlp/seq2seq_loader.py:320-330: features keyed
by the last three characters of the image id, class probabilities, boxes), converts fp16 -> fp32, normalises on the CPU and ships
1.9 MB per sample to the GPU.  That cannot feed ~5000 samples/s per GPU.  Here:

  * `PackedRegionStore`  -- the same arrays packed once into three flat memory-mapped files (fp16 features [N,100,2048], fp16 class
    probabilities [N,100,1601], fp32 boxes [N,100,6]) plus an index; a batch is three fancy-indexed gathers, no per-sample open().
    `pack_from_h5` converts the reference's files (needs h5py, which this image does not ship: it raises a clear error then).
  * `TextPreprocessor`   -- the token half of Preprocess4Seq2seq.__call__ (:229-310) on token ids: truncation, special tokens,
    segment ids, masked-LM corruption with the SAME calls into Python's `random` in the SAME order, so that with an equal seed it
    reproduces the reference's sample stream exactly; it returns lengths instead of the [L, L] mask.
  * `BatchPrefetcher`    -- a background thread that assembles batches into pinned host buffers and copies them to the GPU on a
    separate HIP stream, one batch ahead; it yields the reference's 12-tuple (run_img2txt_dist.py:464) with the compact
    `MaskSpec` / `RawRegions` of vlp_amd.input_prep in the `input_mask` / `vis_pe` slots (the engine expands them on the device).
"""
import json
import os
import queue
import random
import threading

import numpy as np
import torch

from .input_prep import MaskSpec, RawRegions, N_CLS

FEAT_DIM, BOX_DIM = 2048, 6


# ----------------------------------------------------------------------------------------------------
# packed region features
# ----------------------------------------------------------------------------------------------------
def write_packed(out_dir, img_ids, feats, cls_probs, boxes):
    """feats [N,Nv,2048], cls_probs [N,Nv,1601], boxes [N,Nv,6] -> out_dir/{feat.f16,cls.f16,bbox.f32,index.json}."""
    feats, cls_probs, boxes = np.asarray(feats), np.asarray(cls_probs), np.asarray(boxes)
    n, nv = feats.shape[:2]
    if feats.shape != (n, nv, FEAT_DIM) or cls_probs.shape != (n, nv, N_CLS) or boxes.shape != (n, nv, BOX_DIM) or len(img_ids) != n:
        raise ValueError("write_packed: expected feats [N,Nv,2048], cls_probs [N,Nv,1601], boxes [N,Nv,6] and N image ids")
    os.makedirs(out_dir, exist_ok=True)
    feats.astype(np.float16).tofile(os.path.join(out_dir, "feat.f16"))
    cls_probs.astype(np.float16).tofile(os.path.join(out_dir, "cls.f16"))
    boxes.astype(np.float32).tofile(os.path.join(out_dir, "bbox.f32"))
    with open(os.path.join(out_dir, "index.json"), "w") as f:
        json.dump({"num_images": n, "num_regions": nv, "ids": {str(k): i for i, k in enumerate(img_ids)}}, f)


def pack_from_h5(region_det_file_prefix, region_bbox_file, img_ids, out_dir):
    """Converts the reference's layout (seq2seq_loader.py:325-330: `<prefix>_feat<id[-3:]>.h5`, `<prefix>_cls<id[-3:]>.h5`, one bbox
    file, datasets keyed by image id) into a packed store."""
    try:
        import h5py
    except ImportError:
        raise RuntimeError("pack_from_h5 needs h5py to read the reference's feature files; it is not installed in this environment")
    feats, cls_probs, boxes = [], [], []
    with h5py.File(region_bbox_file, "r") as fb:
        for img_id in img_ids:
            with h5py.File(region_det_file_prefix + "_feat" + img_id[-3:] + ".h5", "r") as ff:
                feats.append(ff[img_id][:])
            with h5py.File(region_det_file_prefix + "_cls" + img_id[-3:] + ".h5", "r") as fc:
                cls_probs.append(fc[img_id][:])
            boxes.append(fb[img_id][:])
    write_packed(out_dir, img_ids, np.stack(feats), np.stack(cls_probs), np.stack(boxes))


class PackedRegionStore(object):
    """Memory-mapped view of a packed store; `gather(rows, out...)` copies a batch into caller buffers (pinned host memory)."""

    def __init__(self, path):
        with open(os.path.join(path, "index.json")) as f:
            idx = json.load(f)
        self.n, self.nv = idx["num_images"], idx["num_regions"]
        self.row_of = idx["ids"]
        self.feat = np.memmap(os.path.join(path, "feat.f16"), dtype=np.float16, mode="r", shape=(self.n, self.nv, FEAT_DIM))
        self.cls = np.memmap(os.path.join(path, "cls.f16"), dtype=np.float16, mode="r", shape=(self.n, self.nv, N_CLS))
        self.bbox = np.memmap(os.path.join(path, "bbox.f32"), dtype=np.float32, mode="r", shape=(self.n, self.nv, BOX_DIM))

    def __len__(self):
        return self.n

    def rows(self, img_ids):
        return [self.row_of[str(i)] for i in img_ids]

    def gather(self, rows, feat_out, cls_out, bbox_out):
        """rows: list of store rows; *_out: numpy views (e.g. of pinned torch tensors) shaped [B, Nv, *]."""
        for j, r in enumerate(rows):                 # row-wise memcpy out of the page cache; no temporary
            feat_out[j] = self.feat[r]
            cls_out[j] = self.cls[r]
            bbox_out[j] = self.bbox[r]


# ----------------------------------------------------------------------------------------------------
# text side of Preprocess4Seq2seq.__call__
# ----------------------------------------------------------------------------------------------------
class TextPreprocessor(object):
    """Token half of vlp/seq2seq_loader.py:229-310 on token ids.  `vocab_size` plays the role of len(vocab_words) (the reference
    passes list(tokenizer.vocab.keys()), so a random word's id is the drawn index, :17-19 of loader_utils.py).  Uses the global
    `random` module exactly as the reference does (truncation coin flips, shuffle of the candidate positions, 80/10/10 rule)."""

    def __init__(self, max_pred, mask_prob, vocab_size, cls_id, sep_id, mask_id, unk_id, max_len, max_len_b, mode="s2s", len_vis_input=100,
                 new_segment_ids=True, trunc_seg="b", always_truncate_tail=True):
        assert mode in ("s2s", "bi")
        self.max_pred, self.mask_prob, self.vocab_size = max_pred, mask_prob, vocab_size
        self.cls_id, self.sep_id, self.mask_id, self.unk_id = cls_id, sep_id, mask_id, unk_id
        self.max_len, self.max_len_b, self.mode, self.len_vis_input = max_len, max_len_b, mode, len_vis_input
        self.new_segment_ids, self.trunc_seg, self.always_truncate_tail = new_segment_ids, trunc_seg, always_truncate_tail
        self.task_idx = 3 if mode == "s2s" else 0               # :205-208

    def _truncate(self, a, b):
        """truncate_tokens_pair (:24-59) with max_len = len_vis_input + max_len_b, max_len_a = 0."""
        limit = self.len_vis_input + self.max_len_b
        while len(a) + len(b) > limit:
            if self.max_len_b > 0 and len(b) > self.max_len_b:
                victim = b
            elif self.trunc_seg:
                victim = a if self.trunc_seg == "a" else b
            else:
                victim = a if len(a) > len(b) else b
            if (not self.always_truncate_tail) and random.random() < 0.5:
                del victim[0]
            else:
                victim.pop()

    def __call__(self, token_ids_b):
        a = [self.unk_id] * self.len_vis_input
        b = list(token_ids_b)
        self._truncate(a, b)
        tokens = [self.cls_id] + a + [self.sep_id] + b + [self.sep_id]
        if self.new_segment_ids:
            sa, sb = (4, 5) if self.mode == "s2s" else (0, 1)
        else:
            sa, sb = 0, 1
        segment_ids = [sa] * (len(a) + 2) + [sb] * (len(b) + 1)
        n_pred = min(self.max_pred, max(1, int(round(len(b) * self.mask_prob))))
        cand = [i for i, tk in enumerate(tokens) if i >= len(a) + 2 and tk != self.cls_id]
        random.shuffle(cand)
        masked_pos = cand[:n_pred]
        masked_ids = [tokens[p] for p in masked_pos]
        for p in masked_pos:
            if random.random() < 0.8:
                tokens[p] = self.mask_id
            elif random.random() < 0.5:
                tokens[p] = random.randint(0, self.vocab_size - 1)
        masked_weights = [1] * len(masked_ids)
        pad = self.max_len - len(tokens)
        tokens.extend([0] * pad)
        segment_ids.extend([0] * pad)
        fill = self.max_pred - n_pred
        return {"input_ids": tokens, "segment_ids": segment_ids, "masked_ids": masked_ids + [0] * fill, "masked_pos": masked_pos + [0] * fill,
                "masked_weights": masked_weights + [0] * fill, "len_a": len(a), "len_b": len(b), "is_s2s": self.mode == "s2s",
                "task_idx": self.task_idx}


# ----------------------------------------------------------------------------------------------------
# batches
# ----------------------------------------------------------------------------------------------------
def distributed_sampler_indices(n, world, rank, epoch, seed=0):
    """The index list torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, seed=seed) yields after
    set_epoch(epoch) -- what the reference's loader iterates (run_img2txt_dist.py:295, 455): ONE permutation of the whole dataset per
    epoch, seeded by seed + epoch and identical on every rank, padded by wrapping around to a multiple of `world`, of which rank r
    takes every world-th element.  A rank therefore sees DIFFERENT samples every epoch (a fixed per-rank shard, reshuffled inside
    itself, would train on statistically different batches)."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist()
    total = -(-n // world) * world
    pad = total - n
    if pad:
        idx += idx[:pad] if pad <= len(idx) else (idx * (-(-pad // len(idx))))[:pad]
    return idx[rank:total:world]


class BatchPrefetcher(object):
    """Iterates device-resident batches.  `examples` is a list of (image id, caption token ids); every sample picks the s2s or the
    bidirectional preprocessor with probabilities (s2s_prob, 1 - s2s_prob) like Img2txtDataset.__getitem__ (:162-166).  One batch
    is prepared ahead on a worker thread: host buffers are pinned and the H2D copies run on their own stream, so they overlap the
    training step; the consumer's stream waits on the copy event only."""

    def __init__(self, store, examples, batch_size, proc_s2s, proc_bi=None, s2s_prob=1.0, device=None, steps=None, depth=2, seed=0,
                 vis_mask_prob=0.0, rank=0, world=1):
        """world > 1: `examples` is the WHOLE dataset on every rank and the per-epoch order is DistributedSampler's
        (distributed_sampler_indices; call set_epoch(e) before iterating epoch e like the reference does, :455)."""
        self.store, self.examples, self.B = store, examples, batch_size
        self.rank, self.world, self.epoch = rank, world, 0
        if world > 1 and steps is None:
            steps = -(-len(examples) // world) // batch_size
        # --vis_mask_prob > 0 (mask_image_regions): int(Nv * prob) distinct region positions per sample (seq2seq_loader.py:267-269); their
        # mask columns stay attendable, as in the reference (its :303-304 fills a copy; VLP_BLOCK_MASKED_REGIONS=1 makes the engine block them)
        self.n_vis_masked = int(store.nv * vis_mask_prob)
        self.procs, self.weights = [proc_s2s, proc_bi or proc_s2s], [s2s_prob, 1.0 - s2s_prob]
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.steps = steps if steps is not None else len(examples) // batch_size
        self.depth, self.seed = depth, seed
        self.L, self.P, self.Nv = proc_s2s.max_len, proc_s2s.max_pred, store.nv
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._slots = [self._alloc() for _ in range(depth + 1)]

    def _alloc(self):
        B, L, P, Nv = self.B, self.L, self.P, self.Nv
        pin = lambda *s, dt: torch.empty(*s, dtype=dt).pin_memory()                                  # noqa: E731
        host = {"feat": pin(B, Nv, FEAT_DIM, dt=torch.float16), "cls": pin(B, Nv, N_CLS, dt=torch.float16), "bbox": pin(B, Nv, BOX_DIM, dt=torch.float32),
                "ids": pin(2, B, L, dt=torch.long), "pred": pin(3, B, P, dt=torch.long), "spec": pin(3, B, dt=torch.int32), "task": pin(B, dt=torch.long)}
        if self.n_vis_masked:
            host["vmp"] = pin(B, self.n_vis_masked, dt=torch.long)
        dev = {k: torch.empty_like(v, device=self.device) for k, v in host.items()}
        return host, dev, torch.cuda.Event()

    def _fill(self, slot, batch_examples):
        host, dev, ev = slot
        # The slot's previous H2D copies were only ENQUEUED when it was last filled; they sit on the copy stream, possibly behind a
        # wait on an unfinished training step.  The pinned buffers below are the DMA source: block this (worker) thread until that
        # event has completed on the device before touching them (a never-recorded event returns at once).
        ev.synchronize()
        rows = self.store.rows([e[0] for e in batch_examples])
        self.store.gather(rows, host["feat"].numpy(), host["cls"].numpy(), host["bbox"].numpy())
        for j, (_, toks) in enumerate(batch_examples):
            proc = random.choices(self.procs, weights=self.weights)[0]
            t = proc(toks)
            host["ids"][0, j] = torch.tensor(t["input_ids"])
            host["ids"][1, j] = torch.tensor(t["segment_ids"])
            host["pred"][0, j] = torch.tensor(t["masked_ids"])
            host["pred"][1, j] = torch.tensor(t["masked_pos"])
            host["pred"][2, j] = torch.tensor(t["masked_weights"])
            host["spec"][0, j], host["spec"][1, j], host["spec"][2, j] = t["len_a"] + 2, t["len_a"] + t["len_b"] + 3, int(t["is_s2s"])
            host["task"][j] = t["task_idx"]
            if self.n_vis_masked:
                host["vmp"][j] = torch.tensor(random.sample(range(1, self.Nv + 1), self.n_vis_masked))      # +1 for [CLS] (:269)
        with torch.cuda.stream(self._copy_stream):
            for k in host:
                dev[k].copy_(host[k], non_blocking=True)
            ev.record(self._copy_stream)

    def _batch(self, slot):
        _, d, ev = slot
        torch.cuda.current_stream(self.device).wait_event(ev)
        B = self.B
        spec = MaskSpec(d["spec"][0], d["spec"][1], d["spec"][2], slot[0]["spec"][1].tolist())      # + host lengths (padding-free step)
        raw = RawRegions(d["bbox"], d["cls"])
        is_next = torch.full((B,), -1, dtype=torch.long, device=self.device)
        vis_masked_pos = d["vmp"] if self.n_vis_masked else torch.zeros(B, 0, dtype=torch.long, device=self.device)
        ans = torch.zeros(B, 1, dtype=torch.float16, device=self.device)
        # (input_ids, segment_ids, input_mask, lm_label_ids, masked_pos, masked_weights, is_next, task_idx, img, vis_masked_pos, vis_pe, ans)
        return (d["ids"][0], d["ids"][1], spec, d["pred"][0], d["pred"][1], d["pred"][2], is_next, d["task"], d["feat"], vis_masked_pos, raw, ans)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def epoch_order(self):
        if self.world > 1:
            return distributed_sampler_indices(len(self.examples), self.world, self.rank, self.epoch)
        order = list(range(len(self.examples)))
        random.Random(self.seed).shuffle(order)
        return order

    def __iter__(self):
        order = self.epoch_order()
        q = queue.Queue(maxsize=self.depth)
        free = queue.Queue()
        for s in self._slots:
            free.put(s)

        def worker():
            try:
                for step in range(self.steps):
                    slot = free.get()
                    ex = [self.examples[order[(step * self.B + j) % len(order)]] for j in range(self.B)]
                    self._fill(slot, ex)
                    q.put(slot)
                q.put(None)
            except BaseException as e:                 # surface loader errors in the training thread
                q.put(e)

        th = threading.Thread(target=worker, daemon=True)
        th.start()
        prev = None
        while True:
            item = q.get()
            if prev is not None:
                # the batch handed out last iteration has been consumed by launches already enqueued; its device buffers may be
                # overwritten once those launches are done: make the copy stream wait for the consumer before recycling the slot
                self._copy_stream.wait_stream(torch.cuda.current_stream(self.device))
                free.put(prev)
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            prev = item
            yield self._batch(item)
        th.join()
