"""Synthetic batches that follow the reference's input contract.

The reference builds every training batch on the CPU in ``vlp/seq2seq_loader.py``
(``Preprocess4Seq2seq.__call__``, lines 229-359) and collates it with
``vlp/loader_utils.py:17-24``; the train loop unpacks it in this order
(``vlp/run_img2txt_dist.py:464``)::

    input_ids, segment_ids, input_mask, lm_label_ids, masked_pos, masked_weights,
    is_next, task_idx, img, vis_masked_pos, vis_pe, ans_labels

There is no dataset on the build or bench machines, so this module restates the *shape and
semantics* of that tuple (token layout ``:238``, segment ids ``:240-246``, the [L, L]
self-attention mask ``:292-304``, masked-LM targets ``:250-311``) and fills it with seeded
random content.  It is host-side plumbing shared by bench.py, the entry script's
``--synthetic`` mode and the tests.
"""
from collections import namedtuple

import torch

# bert-base-cased ids used by the reference (run_img2txt_dist.py:316-317)
PAD_ID, UNK_ID, CLS_ID, SEP_ID, MASK_ID = 0, 100, 101, 102, 103

Batch = namedtuple("Batch", [
    "input_ids", "segment_ids", "input_mask", "lm_label_ids", "masked_pos", "masked_weights",
    "is_next", "task_idx", "img", "vis_masked_pos", "vis_pe", "ans_labels"])


def seq_len(max_len_b, len_vis_input):
    """run_img2txt_dist.py:193 -- +3 for [CLS] and 2x[SEP]."""
    return max_len_b + len_vis_input + 3


def build_attention_mask(L, n_a, n_b, mode):
    """seq2seq_loader.py:292-301.  n_a = len(tokens_a) (regions), n_b = len(tokens_b)."""
    second_st, second_end = n_a + 2, n_a + n_b + 3
    if mode == "s2s":
        m = torch.zeros(L, L, dtype=torch.long)
        m[:, :n_a + 2] = 1
        n = second_end - second_st
        m[second_st:second_end, second_st:second_end] = torch.tril(torch.ones(n, n, dtype=torch.long))
    elif mode == "bi":
        row = torch.zeros(L, dtype=torch.long)
        row[:second_end] = 1
        m = row.unsqueeze(0).expand(L, L).clone()
    else:
        raise ValueError("mode must be 's2s' or 'bi'")
    return m


def make_batch(batch_size, max_len_b=64, len_vis_input=100, vocab_size=28996, max_pred=3,
               mask_prob=0.15, s2s_prob=1.0, tasks="img2txt", seed=1234, new_segment_ids=True,
               feat_dim=2048, pe_dim=1607, num_answers=3129, dtype=torch.float32, min_len_b=6, vis_mask_prob=0.0,
               block_masked_regions=False):
    """Returns a Batch of CPU tensors (float tensors in ``dtype``).  vis_mask_prob > 0 (--vis_mask_prob, mask_image_regions): per sample
    int(len_vis_input * vis_mask_prob) distinct region positions in 1..len_vis_input (seq2seq_loader.py:267-269).  The reference's
    `input_mask[:, vis_masked_pos].fill_(0)` (:303-304) indexes with a numpy array, fills a copy and leaves the mask as it was, so by
    default the mask is NOT edited here either; block_masked_regions=True applies what that line's comment intends."""
    g = torch.Generator().manual_seed(seed)
    B, Nv = batch_size, len_vis_input
    L = seq_len(max_len_b, Nv)
    lo_tok = min(1000, vocab_size // 2)

    input_ids = torch.zeros(B, L, dtype=torch.long)
    segment_ids = torch.zeros(B, L, dtype=torch.long)
    input_mask = torch.zeros(B, L, L, dtype=torch.long)
    lm_label_ids = torch.zeros(B, max_pred, dtype=torch.long)
    masked_pos = torch.zeros(B, max_pred, dtype=torch.long)
    masked_weights = torch.zeros(B, max_pred, dtype=torch.long)
    task_idx = torch.zeros(B, dtype=torch.long)

    min_len_b = min(min_len_b, max_len_b)
    for b in range(B):
        n_b = int(torch.randint(min_len_b, max_len_b + 1, (1,), generator=g))
        mode = "s2s" if float(torch.rand(1, generator=g)) < s2s_prob else "bi"
        if tasks == "vqa2":
            mode = "bi"
        toks = torch.randint(lo_tok, vocab_size, (n_b,), generator=g)
        ids = [CLS_ID] + [UNK_ID] * Nv + [SEP_ID] + toks.tolist() + [SEP_ID]
        if new_segment_ids:
            seg = ([4] * (Nv + 2) + [5] * (n_b + 1)) if mode == "s2s" else ([0] * (Nv + 2) + [1] * (n_b + 1))
        else:
            seg = [0] * (Nv + 2) + [1] * (n_b + 1)
        # masked-LM targets: only tokens_b and the closing [SEP] are candidates (:256-265)
        n_pred = min(max_pred, max(1, int(round(n_b * mask_prob))))
        cand = torch.arange(Nv + 2, Nv + 2 + n_b + 1)
        perm = torch.randperm(cand.numel(), generator=g)[:n_pred]
        pos = cand[perm].tolist()
        labels = [ids[p] for p in pos]
        for p in pos:
            r = float(torch.rand(1, generator=g))
            if r < 0.8:
                ids[p] = MASK_ID
            elif r < 0.9:
                ids[p] = int(torch.randint(lo_tok, vocab_size, (1,), generator=g))
        input_ids[b, :len(ids)] = torch.tensor(ids)
        segment_ids[b, :len(seg)] = torch.tensor(seg)
        input_mask[b] = build_attention_mask(L, Nv, n_b, mode)
        lm_label_ids[b, :n_pred] = torch.tensor(labels)
        masked_pos[b, :n_pred] = torch.tensor(pos)
        masked_weights[b, :n_pred] = 1
        task_idx[b] = 3 if mode == "s2s" else 0

    # Detectron fc6 region features are post-ReLU (non-negative)
    img = torch.randn(B, Nv, feat_dim, generator=g).abs_()
    # seq2seq_loader.py:348-351: cat(layer_norm(box6), layer_norm(cls_prob 1601))
    n_box = 6
    box = torch.rand(B, Nv, n_box, generator=g)
    cls = torch.softmax(torch.randn(B, Nv, pe_dim - n_box, generator=g), dim=-1)
    vis_pe = torch.cat((torch.nn.functional.layer_norm(box, [n_box]),
                        torch.nn.functional.layer_norm(cls, [pe_dim - n_box])), dim=-1)
    if tasks == "vqa2":
        dens = (torch.rand(B, num_answers, generator=g) < (10.0 / num_answers)).float()
        ans_labels = torch.rand(B, num_answers, generator=g) * dens
    else:
        ans_labels = torch.zeros(B, 1)
    is_next = torch.full((B,), -1, dtype=torch.long)
    n_vm = int(Nv * vis_mask_prob)
    if n_vm > 0:
        vis_masked_pos = torch.stack([torch.randperm(Nv, generator=g)[:n_vm] + 1 for _ in range(B)])     # +1 for [CLS] (:269)
        if block_masked_regions:
            for b in range(B):
                input_mask[b][:, vis_masked_pos[b]] = 0                                                 # what the comment of :304 intends
    else:
        vis_masked_pos = torch.zeros(B, 0, dtype=torch.long)
    return Batch(input_ids, segment_ids, input_mask, lm_label_ids, masked_pos, masked_weights,
                 is_next, task_idx, img.to(dtype), vis_masked_pos, vis_pe.to(dtype), ans_labels.to(dtype))


def batch_to(batch, device, half=False):
    """run_img2txt_dist.py:463-468: everything .to(device); img / vis_pe .half() under --fp16."""
    out = []
    for name, t in zip(Batch._fields, batch):
        t = t.to(device)
        if half and name in ("img", "vis_pe"):
            t = t.half()
        out.append(t)
    return Batch(*out)
