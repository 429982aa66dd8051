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
  * `BatchPrefetcher`    -- `num_workers` background threads (the reference's DataLoader `num_workers`, run_img2txt_dist.py:296-298),
    each assembling WHOLE batches into pinned host buffers (one `np.take(..., out=)` per feature array straight out of the page cache --
    numpy releases the GIL for it, so the threads scale -- and one ndarray write per token field) and copying them to the GPU on a
    separate HIP stream, `depth` batches ahead; it yields the reference's 12-tuple (run_img2txt_dist.py:464) with the compact
    `MaskSpec` / `RawRegions` of vlp_amd.input_prep in the `input_mask` / `vis_pe` slots (the engine expands them on the device).
    Every batch draws from its OWN `random.Random(batch_seed(seed, epoch, rank, step))`: its content is a pure function of those four
    numbers -- the same for 1 and K workers, independent of thread interleaving and of anything else that touches the global `random`.
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
        """rows: list of store rows; *_out: numpy views (e.g. of pinned torch tensors) shaped [B, Nv, *].  One np.take per array straight
        from the memory map into the caller's buffer (no temporary; mode='clip' because mode='raise' buffers `out` -- the rows are checked
        here instead); numpy drops the GIL for the copy, so several loader threads gather concurrently."""
        rows = np.asarray(rows, dtype=np.intp)
        if rows.size and (rows.min() < 0 or rows.max() >= self.n):
            raise IndexError("PackedRegionStore.gather: row out of range [0, %d)" % self.n)
        np.take(self.feat, rows, axis=0, out=feat_out, mode="clip")
        np.take(self.cls, rows, axis=0, out=cls_out, mode="clip")
        np.take(self.bbox, rows, axis=0, out=bbox_out, mode="clip")


# ----------------------------------------------------------------------------------------------------
# text side of Preprocess4Seq2seq.__call__
# ----------------------------------------------------------------------------------------------------
class TextPreprocessor(object):
    """Token half of vlp/seq2seq_loader.py:229-310 on token ids.  `vocab_size` plays the role of len(vocab_words) (the reference
    passes list(tokenizer.vocab.keys()), so a random word's id is the drawn index, :17-19 of loader_utils.py).  Draws from `rng` -- by
    default the global `random` module, exactly as the reference does (truncation coin flips, shuffle of the candidate positions,
    80/10/10 rule: same seed, same sample stream, pinned against the unmodified Preprocess4Seq2seq); the prefetcher passes a private
    `random.Random` per batch so that loader threads neither race on the global state nor depend on their interleaving."""

    def __init__(self, max_pred, mask_prob, vocab_size, cls_id, sep_id, mask_id, unk_id, max_len, max_len_b, mode="s2s", len_vis_input=100,
                 new_segment_ids=True, trunc_seg="b", always_truncate_tail=True):
        assert mode in ("s2s", "bi")
        self.max_pred, self.mask_prob, self.vocab_size = max_pred, mask_prob, vocab_size
        self.cls_id, self.sep_id, self.mask_id, self.unk_id = cls_id, sep_id, mask_id, unk_id
        self.max_len, self.max_len_b, self.mode, self.len_vis_input = max_len, max_len_b, mode, len_vis_input
        self.new_segment_ids, self.trunc_seg, self.always_truncate_tail = new_segment_ids, trunc_seg, always_truncate_tail
        self.task_idx = 3 if mode == "s2s" else 0               # :205-208

    def _truncate(self, a, b, rng=random):
        """truncate_tokens_pair (:24-59) with max_len = len_vis_input + max_len_b, max_len_a = 0."""
        limit = self.len_vis_input + self.max_len_b
        while len(a) + len(b) > limit:
            if self.max_len_b > 0 and len(b) > self.max_len_b:
                victim = b
            elif self.trunc_seg:
                victim = a if self.trunc_seg == "a" else b
            else:
                victim = a if len(a) > len(b) else b
            if (not self.always_truncate_tail) and rng.random() < 0.5:
                del victim[0]
            else:
                victim.pop()

    def __call__(self, token_ids_b, rng=random):
        a = [self.unk_id] * self.len_vis_input
        b = list(token_ids_b)
        self._truncate(a, b, rng)
        tokens = [self.cls_id] + a + [self.sep_id] + b + [self.sep_id]
        if self.new_segment_ids:
            sa, sb = (4, 5) if self.mode == "s2s" else (0, 1)
        else:
            sa, sb = 0, 1
        segment_ids = [sa] * (len(a) + 2) + [sb] * (len(b) + 1)
        n_pred = min(self.max_pred, max(1, int(round(len(b) * self.mask_prob))))
        cand = [i for i, tk in enumerate(tokens) if i >= len(a) + 2 and tk != self.cls_id]
        rng.shuffle(cand)
        masked_pos = cand[:n_pred]
        masked_ids = [tokens[p] for p in masked_pos]
        for p in masked_pos:
            if rng.random() < 0.8:
                tokens[p] = self.mask_id
            elif rng.random() < 0.5:
                tokens[p] = rng.randint(0, self.vocab_size - 1)
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


def balanced_rank_split(lengths, world, cap):
    """Deal len(lengths) = cap * world samples to `world` ranks, exactly `cap` each, so that the per-rank sums of `lengths` are as equal
    as a greedy deal gets them: longest sample first, each to the rank with the smallest sum that still has room (ties: lowest rank;
    equal lengths keep their input order).  Deterministic, so every rank computes the same deal.  Returns one list of positions
    (indices into `lengths`) per rank."""
    n = len(lengths)
    assert n == cap * world
    order = sorted(range(n), key=lambda i: (-lengths[i], i))
    sums, parts = [0] * world, [[] for _ in range(world)]
    for i in order:
        r = min((r_ for r_ in range(world) if len(parts[r_]) < cap), key=lambda r_: (sums[r_], r_))
        parts[r].append(i)
        sums[r] += lengths[i]
    return parts


def balanced_epoch_order(n, world, rank, epoch, batch_size, length_of, seed=0):
    """Length-balanced variant of distributed_sampler_indices for padding-free steps (each rank's GEMMs run over sum(kept lengths) rows,
    and the gradient all-reduce makes every step as slow as the rank with the most rows).  The epoch's permutation and its padding are
    DistributedSampler's; global batch s = permutation[s * batch_size * world : (s + 1) * batch_size * world] is exactly the set of
    samples the `world` ranks of the reference see in step s (rank r takes every world-th element of it).  Here that same set is dealt
    to the ranks by kept length (balanced_rank_split) instead of by position: same samples per optimizer step, same mean gradient up to
    summation order, near-equal rows per rank.  `length_of(i)` = kept rows of dataset index i."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist()
    total = -(-n // world) * world
    pad = total - n
    if pad:
        idx += idx[:pad] if pad <= len(idx) else (idx * (-(-pad // len(idx))))[:pad]
    per_rank = total // world
    out = []
    for s0 in range(0, per_rank, batch_size):
        cap = min(batch_size, per_rank - s0)
        block = idx[s0 * world:(s0 + cap) * world]
        parts = balanced_rank_split([length_of(i) for i in block], world, cap)
        out.extend(block[j] for j in parts[rank])
    return out


def batch_seed(seed, epoch, rank, step):
    """Seed of the private `random.Random` of batch `step` (of `epoch`, on `rank`): a fixed integer mix, so the content of a batch does
    not depend on how many loader threads there are or which one filled it."""
    x = (int(seed) * 0x9E3779B97F4A7C15 + int(epoch) * 0xD1B54A32D192ED03 + int(rank) * 0x8CB92BA72F3D8DD7 + int(step) * 0xBF58476D1CE4E5B9 + 0x632BE59BD9B4E019)
    x &= (1 << 64) - 1
    x ^= x >> 31
    return x


class BatchPrefetcher(object):
    """Iterates device-resident batches.  `examples` is a list of (image id, caption token ids); every sample picks the s2s or the
    bidirectional preprocessor with probabilities (s2s_prob, 1 - s2s_prob) like Img2txtDataset.__getitem__ (:162-166).  `num_workers`
    threads (run_img2txt_dist.py:296-298 `--num_workers`) prepare up to `depth` batches ahead: host buffers are pinned and the H2D
    copies run on their own stream, so they overlap the training step; the consumer's stream waits on the copy event only.  Batches are
    handed out in step order whatever the workers' completion order, and batch `step` draws from random.Random(batch_seed(seed, epoch,
    rank, step)) only: the same seed gives the same batches for any worker count (tests/test_data_cpu.py)."""

    def __init__(self, store, examples, batch_size, proc_s2s, proc_bi=None, s2s_prob=1.0, device=None, steps=None, depth=None, seed=0,
                 vis_mask_prob=0.0, rank=0, world=1, num_workers=1, balance_lengths=False):
        """world > 1: `examples` is the WHOLE dataset on every rank and the per-epoch order is DistributedSampler's
        (distributed_sampler_indices; call set_epoch(e) before iterating epoch e like the reference does, :455).
        steps: batches per epoch; default ceil(samples of this rank / batch_size) = len(DataLoader) of the reference (drop_last=False,
        :296-298) -- the last batch is filled by wrapping around the epoch's order instead of being short (fixed shapes).
        balance_lengths (world > 1): every global batch of batch_size x world samples is dealt to the ranks by kept length
        (balanced_rank_split) instead of by index -- same sample SET per step as DistributedSampler, near-equal padding-free row counts."""
        self.store, self.examples, self.B = store, examples, batch_size
        self.rank, self.world, self.epoch = rank, world, 0
        self.balance_lengths = bool(balance_lengths) and world > 1
        per_rank = -(-len(examples) // world)
        if steps is None:
            steps = -(-per_rank // batch_size)
        if steps <= 0:
            raise ValueError("BatchPrefetcher: no steps (%d examples, world %d, batch %d)" % (len(examples), world, batch_size))
        # --vis_mask_prob > 0 (mask_image_regions): int(Nv * prob) distinct region positions per sample (seq2seq_loader.py:267-269); their
        # mask columns stay attendable, as in the reference (its :303-304 fills a copy; VLP_BLOCK_MASKED_REGIONS=1 makes the engine block them)
        self.n_vis_masked = int(store.nv * vis_mask_prob)
        self.procs, self.weights = [proc_s2s, proc_bi or proc_s2s], [s2s_prob, 1.0 - s2s_prob]
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.steps = steps
        self.num_workers = max(1, int(num_workers))
        self.depth = max(2, self.num_workers) if depth is None else max(1, int(depth))
        self.seed = seed
        self.L, self.P, self.Nv = proc_s2s.max_len, proc_s2s.max_pred, store.nv
        self._copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._slots = [self._alloc() for _ in range(self.depth + 1)]

    def _alloc(self):
        B, L, P, Nv = self.B, self.L, self.P, self.Nv
        cuda = self._copy_stream is not None
        pin = (lambda *s, dt: torch.empty(*s, dtype=dt).pin_memory()) if cuda else (lambda *s, dt: torch.empty(*s, dtype=dt))      # noqa: E731
        host = {"feat": pin(B, Nv, FEAT_DIM, dt=torch.float16), "cls": pin(B, Nv, N_CLS, dt=torch.float16), "bbox": pin(B, Nv, BOX_DIM, dt=torch.float32),
                "ids": pin(2, B, L, dt=torch.long), "pred": pin(3, B, P, dt=torch.long), "spec": pin(3, B, dt=torch.int32), "task": pin(B, dt=torch.long)}
        if self.n_vis_masked:
            host["vmp"] = pin(B, self.n_vis_masked, dt=torch.long)
        dev = {k: torch.empty_like(v, device=self.device) for k, v in host.items()}
        # numpy views of the host buffers, taken once (the fill writes whole arrays through them)
        views = {k: v.numpy() for k, v in host.items()}
        return host, dev, (torch.cuda.Event() if cuda else None), views

    def _fill(self, slot, batch_examples, rng):
        host, dev, ev, hv = slot
        # The slot's previous H2D copies were only ENQUEUED when it was last filled; they sit on the copy stream, possibly behind a
        # wait on an unfinished training step.  The pinned buffers below are the DMA source: block this (worker) thread until that
        # event has completed on the device before touching them (a never-recorded event returns at once).
        if ev is not None:
            ev.synchronize()
        rows = self.store.rows([e[0] for e in batch_examples])
        self.store.gather(rows, hv["feat"], hv["cls"], hv["bbox"])
        # token side: python lists per sample (the reference's own arithmetic, on this batch's private generator), ONE ndarray write per field
        B = len(batch_examples)
        ids, seg, mid, mpos, mw = [None] * B, [None] * B, [None] * B, [None] * B, [None] * B
        spec = np.empty((3, B), dtype=np.int32)
        task = np.empty((B,), dtype=np.int64)
        vmp = [None] * B
        for j, (_, toks) in enumerate(batch_examples):
            proc = rng.choices(self.procs, weights=self.weights)[0]
            t = proc(toks, rng)
            ids[j], seg[j], mid[j], mpos[j], mw[j] = t["input_ids"], t["segment_ids"], t["masked_ids"], t["masked_pos"], t["masked_weights"]
            spec[0, j], spec[1, j], spec[2, j] = t["len_a"] + 2, t["len_a"] + t["len_b"] + 3, int(t["is_s2s"])
            task[j] = t["task_idx"]
            if self.n_vis_masked:
                vmp[j] = rng.sample(range(1, self.Nv + 1), self.n_vis_masked)      # +1 for [CLS] (:269)
        hv["ids"][0], hv["ids"][1] = np.asarray(ids, dtype=np.int64), np.asarray(seg, dtype=np.int64)
        hv["pred"][0], hv["pred"][1], hv["pred"][2] = np.asarray(mid, dtype=np.int64), np.asarray(mpos, dtype=np.int64), np.asarray(mw, dtype=np.int64)
        hv["spec"][...] = spec
        hv["task"][...] = task
        if self.n_vis_masked:
            hv["vmp"][...] = np.asarray(vmp, dtype=np.int64)
        lens_host = spec[1].tolist()              # second_end on the host: the padding-free step needs no device read-back
        if ev is None:
            for k in host:
                dev[k].copy_(host[k])
            return lens_host
        with torch.cuda.stream(self._copy_stream):
            for k in host:
                dev[k].copy_(host[k], non_blocking=True)
            ev.record(self._copy_stream)
        return lens_host

    def _batch(self, slot, lens_host):
        _, d, ev, _ = slot
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        B = self.B
        spec = MaskSpec(d["spec"][0], d["spec"][1], d["spec"][2], lens_host)      # + host lengths (padding-free step)
        raw = RawRegions(d["bbox"], d["cls"])
        is_next = torch.full((B,), -1, dtype=torch.long, device=self.device)
        vis_masked_pos = d["vmp"] if self.n_vis_masked else torch.zeros(B, 0, dtype=torch.long, device=self.device)
        ans = torch.zeros(B, 1, dtype=torch.float16, device=self.device)
        # (input_ids, segment_ids, input_mask, lm_label_ids, masked_pos, masked_weights, is_next, task_idx, img, vis_masked_pos, vis_pe, ans)
        return (d["ids"][0], d["ids"][1], spec, d["pred"][0], d["pred"][1], d["pred"][2], is_next, d["task"], d["feat"], vis_masked_pos, raw, ans)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def epoch_order(self):
        """Indices into `examples` this rank walks in the current epoch, `steps * batch_size` of them at most are used."""
        if self.world > 1:
            if self.balance_lengths:
                return balanced_epoch_order(len(self.examples), self.world, self.rank, self.epoch, self.B, self._kept_length)
            return distributed_sampler_indices(len(self.examples), self.world, self.rank, self.epoch)
        order = list(range(len(self.examples)))
        random.Random(batch_seed(self.seed, self.epoch, 0, -1)).shuffle(order)        # RandomSampler(replacement=False): a fresh permutation per epoch (:292)
        return order

    def _kept_length(self, i):
        """Rows sample i occupies in a padding-free step: [CLS] + regions + [SEP] + its (truncated) caption + [SEP]."""
        p = self.procs[0]
        return p.len_vis_input + 3 + min(len(self.examples[i][1]), p.max_len_b)

    def step_examples(self, order, step):
        return [self.examples[order[(step * self.B + j) % len(order)]] for j in range(self.B)]

    def __iter__(self):
        order = self.epoch_order()
        epoch = self.epoch
        free = queue.Queue()
        for s in self._slots:
            free.put(s)
        results = {}                       # step -> (slot, lens_host) | BaseException
        cond = threading.Condition()
        next_step = [0]
        stop = [False]

        def worker():
            while True:
                # slot first, step second: a claimed step always owns a slot, so the lowest outstanding step (the one the consumer is
                # waiting for) can never starve behind younger steps that took the slots
                slot = free.get()
                if slot is None:           # shutdown
                    return
                with cond:
                    if stop[0] or next_step[0] >= self.steps:
                        free.put(slot)
                        return
                    step = next_step[0]
                    next_step[0] += 1
                try:
                    lens = self._fill(slot, self.step_examples(order, step), random.Random(batch_seed(self.seed, epoch, self.rank, step)))
                    item = (slot, lens)
                except BaseException as e:                 # surface loader errors in the training thread
                    item = e
                with cond:
                    results[step] = item
                    cond.notify_all()

        threads = [threading.Thread(target=worker, daemon=True) for _ in range(min(self.num_workers, self.steps))]
        for th in threads:
            th.start()
        prev = None
        try:
            for step in range(self.steps):
                with cond:
                    while step not in results:
                        cond.wait()
                    item = results.pop(step)
                if prev is not None:
                    # the batch handed out last iteration has been consumed by launches already enqueued; its device buffers may be
                    # overwritten once those launches are done: make the copy stream wait for the consumer before recycling the slot
                    if self._copy_stream is not None:
                        self._copy_stream.wait_stream(torch.cuda.current_stream(self.device))
                    free.put(prev)
                    prev = None
                if isinstance(item, BaseException):
                    raise item
                prev = item[0]
                yield self._batch(*item)
        finally:
            with cond:
                stop[0] = True
            for _ in threads:
                free.put(None)             # wake workers blocked on a free slot
            for th in threads:
                th.join(timeout=10.0)
