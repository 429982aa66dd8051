"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the UNMODIFIED reference.

Run in the build container (needs /root/reference):   python -m oracle.make_golden

For each case: parameters are drawn with oracle.vlp_oracle.init_params(seed) (reproducible from the
seed alone, so the multi-MB weights never need to be committed), loaded into the reference's own
BertForPreTrainingLossMask (strict state_dict load => also pins checkpoint key names), a seeded
synthetic batch (vlp_amd.synthetic) is pushed through the reference forward + backward, and the
reference's outputs are stored: losses, MLM / VQA logits (captured by forward hooks on
`cls.predictions` / `ans_classifier`, modeling.py:1102,1139 -- the reference never returns logits),
a strided sample of every layer's hidden states, per-parameter gradient L2 norms and strided samples
of selected gradients, and the parameters after one reference BertAdam.step().
A fingerprint of the generated parameters/batch is stored too so a consumer can tell an RNG-stream
mismatch from a parity failure.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_loader, vlp_oracle as O          # noqa: E402
from vlp_amd import synthetic as S                      # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    # name: (model kwargs, batch kwargs)
    "img2txt_L123_2l": (dict(vocab_size=1024, layers=2, tasks="img2txt", seed=11),
                        dict(batch_size=2, max_len_b=20, vocab_size=1024, max_pred=3, s2s_prob=0.5, seed=101)),
    "img2txt_L167_2l": (dict(vocab_size=1024, layers=2, tasks="img2txt", seed=12),
                        dict(batch_size=3, max_len_b=64, vocab_size=1024, max_pred=3, s2s_prob=1.0, seed=102)),
    "vqa2_L123_2l": (dict(vocab_size=1024, layers=2, tasks="vqa2", seed=13),
                     dict(batch_size=2, max_len_b=20, vocab_size=1024, max_pred=1, tasks="vqa2", seed=103)),
    "img2txt_L167_12l": (dict(vocab_size=2048, layers=12, tasks="img2txt", seed=14),
                         dict(batch_size=2, max_len_b=64, vocab_size=2048, max_pred=3, s2s_prob=0.75, seed=104)),
    # --vis_mask_prob 0.25: masked region rows + the Selfie-style pretext loss over the pooled output (modeling.py:1049-1056, 1113-1131);
    # the reference's .byte() row mask (:1050) is shimmed to .bool() around the unmodified call (torch >= 1.2 refuses byte masks)
    "img2txt_L123_2l_vismask": (dict(vocab_size=1024, layers=2, tasks="img2txt", seed=15, mask_image_regions=True),
                                dict(batch_size=3, max_len_b=20, vocab_size=1024, max_pred=3, s2s_prob=0.5, seed=105, vis_mask_prob=0.25)),
    "vqa2_L123_2l_vismask": (dict(vocab_size=1024, layers=2, tasks="vqa2", seed=16, mask_image_regions=True),
                             dict(batch_size=2, max_len_b=20, vocab_size=1024, max_pred=1, tasks="vqa2", seed=106, vis_mask_prob=0.25)),
}

# greedy decoding cases (BertForSeq2SeqDecoder, modeling.py:1189-1253): (model kwargs, B, T, input seed)
DECODE_CASES = {
    "decode_2l_T12": (dict(vocab_size=1024, layers=2, tasks="img2txt", seed=21, std=0.05), 3, 12, 204),
    "decode_12l_T20": (dict(vocab_size=2048, layers=12, tasks="img2txt", seed=22, std=0.04), 2, 20, 202),
}

# beam search cases (BertForSeq2SeqDecoder.beam_search, modeling.py:1255-1494): (model kwargs, B, T, input seed, decoder kwargs)
BEAM_CASES = {
    "beam_2l_K3": (dict(vocab_size=1024, layers=2, tasks="img2txt", seed=27, std=0.1), 3, 10, 213,
                   dict(search_beam_size=3, length_penalty=0.4, min_len=2, forbid_duplicate_ngrams=True, ngram_size=2)),
    "beam_12l_K5": (dict(vocab_size=2048, layers=12, tasks="img2txt", seed=24, std=0.04), 2, 12, 212,
                    dict(search_beam_size=5, length_penalty=1.0, min_len=0, forbid_duplicate_ngrams=False, ngram_size=3)),
}

# input-preparation cases (Preprocess4Seq2seq.__call__, vlp/seq2seq_loader.py:229-359): (mode, caption tokens, seed)
LOADER_CASES = {"loader_s2s_n9": ("s2s", 9, 1), "loader_bi_n14": ("bi", 14, 2)}

GRAD_SAMPLES = [
    "bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
    "bert.embeddings.token_type_embeddings.weight", "bert.embeddings.LayerNorm.weight",
    "bert.encoder.layer.0.attention.self.query.weight", "bert.encoder.layer.0.attention.self.value.bias",
    "bert.encoder.layer.1.intermediate.dense.weight", "bert.encoder.layer.1.output.LayerNorm.bias",
    "vis_embed.0.weight", "vis_embed.2.bias", "vis_pe_embed.0.weight",
    "cls.predictions.transform.dense.weight", "cls.predictions.bias", "ans_classifier.2.weight",
    "bert.pooler.dense.weight", "bert.pooler.dense.bias",      # live in the vismask cases only
]


def sample(t, n=4096):
    """Deterministic strided sample of a tensor (flattened)."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].float().numpy().copy()


def fingerprint(p, batch):
    keys = sorted(p.keys())
    fp = [float(p[k].double().abs().sum()) for k in keys[:8]]
    fp += [float(batch.img.double().sum()), float(batch.vis_pe.double().abs().sum()),
           float(batch.input_ids.sum()), float(batch.input_mask.sum())]
    return np.asarray(fp, dtype=np.float64)


def run_reference_case(mk, bk):
    ref = ref_loader.load_reference()
    tasks = mk["tasks"]
    p = O.init_params(vocab_size=mk["vocab_size"], layers=mk["layers"], tasks=tasks, seed=mk["seed"])
    model = ref_loader.build_reference_model(dict(vocab_size=mk["vocab_size"], num_hidden_layers=mk["layers"]),
                                             tasks=tasks, seed=0)
    sd = dict(p)
    sd["cls.predictions.decoder.weight"] = p["bert.embeddings.word_embeddings.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    model.eval()
    batch = S.make_batch(**bk)
    cap = {}
    model.cls.predictions.register_forward_hook(lambda m, i, o: cap.__setitem__("mlm_logits", o.detach()))
    if tasks == "vqa2":
        model.ans_classifier.register_forward_hook(lambda m, i, o: cap.__setitem__("vqa_logits", o.detach()))
    hid = []
    model.bert.embeddings.register_forward_hook(lambda m, i, o: hid.append(o.detach()))
    for lyr in model.bert.encoder.layer:
        lyr.register_forward_hook(lambda m, i, o: hid.append(o.detach()))
    mir = bool(mk.get("mask_image_regions", False))
    true_byte = torch.Tensor.byte
    if mir:
        torch.Tensor.byte = lambda self: self.bool()
        model.bert.pooler.register_forward_hook(lambda m, i, o: cap.__setitem__("pooled_output", o.detach()))
    try:
        losses = model(batch.img, batch.vis_pe, batch.input_ids, batch.segment_ids, batch.input_mask,
                       batch.lm_label_ids, batch.ans_labels, batch.is_next, masked_pos=batch.masked_pos,
                       masked_weights=batch.masked_weights, task_idx=batch.task_idx,
                       vis_masked_pos=batch.vis_masked_pos, mask_image_regions=mir, drop_worst_ratio=0)
    finally:
        torch.Tensor.byte = true_byte
    loss = losses[0] + losses[1] + losses[2]          # run_img2txt_dist.py:531
    loss.sum().backward()
    out = {"fingerprint": fingerprint(p, batch),
           "loss_shapes": np.asarray([l.dim() for l in losses]),
           "losses": np.asarray([float(l.detach().sum()) for l in losses], dtype=np.float64)}
    for k, v in cap.items():
        out[k] = v.float().numpy()
    for i, h in enumerate(hid):
        out["hidden_%d" % i] = sample(h)
    names, norms = [], []
    for n, q in model.named_parameters():
        names.append(n)
        norms.append(-1.0 if q.grad is None else float(q.grad.double().norm()))
        if n in GRAD_SAMPLES and q.grad is not None:
            out["grad::" + n] = sample(q.grad)
    out["param_names"] = np.asarray(names)
    out["grad_norms"] = np.asarray(norms, dtype=np.float64)
    # one reference BertAdam step (optimization.py:112-182) with the train script's param groups
    # (run_img2txt_dist.py:394-401,422-426)
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    named = list(model.named_parameters())
    groups = [{"params": [q for n, q in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [q for n, q in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = ref.optimization.BertAdam(groups, lr=1e-2, warmup=0.1, schedule="warmup_linear", t_total=20)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        opt.step()          # step 0 of a warmup schedule has lr 0 ...
        opt.step()          # ... so take two steps with the same gradients
    for n, q in named:
        if n in GRAD_SAMPLES:
            out["adam::" + n] = sample(q.detach() - p[n])      # the update, not the weight
    return out


def decode_inputs(B, T, seed, Nv=100):
    """Seeded inputs of a greedy decoding call, shaped like seq2seq_loader.Preprocess4Seq2seqDecoder's output
    (seq2seq_loader.py:383-486): [CLS] + Nv placeholders + [SEP] as input_ids, segment 4 | 5, causal mask on the target part."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, Nv, 2048, generator=g).abs()
    vis_pe = torch.rand(B, Nv, 1607, generator=g)
    in_len, out_len = Nv + 2, Nv + 2 + T
    input_ids = torch.tensor([[S.CLS_ID] + [S.UNK_ID] * Nv + [S.SEP_ID]] * B)
    token_type = torch.tensor([[4] * in_len + [5] * T] * B)
    pos = torch.arange(out_len).unsqueeze(0).expand(B, -1).contiguous()
    am = torch.zeros(B, out_len, out_len, dtype=torch.long)
    am[:, :, :in_len] = 1
    am[:, in_len:, in_len:] = torch.tril(torch.ones(T, T, dtype=torch.long))
    return img, vis_pe, input_ids, token_type, pos, am


def decode_fingerprint(p, inp):
    keys = sorted(p.keys())
    fp = [float(p[k].double().abs().sum()) for k in keys[:8]]
    fp += [float(inp[0].double().sum()), float(inp[1].double().sum())]
    return np.asarray(fp, dtype=np.float64)


def run_reference_decode_case(mk, B, T, seed):
    p = O.init_params(vocab_size=mk["vocab_size"], layers=mk["layers"], tasks=mk["tasks"], seed=mk["seed"], std=mk["std"])
    dec = ref_loader.build_reference_model(dict(vocab_size=mk["vocab_size"], num_hidden_layers=mk["layers"]), seed=0, decoder=True,
                                           mask_word_id=S.MASK_ID, eos_id=S.SEP_ID)
    sd = dict(p)
    sd["cls.predictions.decoder.weight"] = p["bert.embeddings.word_embeddings.weight"]
    missing, unexpected = dec.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    dec.eval()
    inp = decode_inputs(B, T, seed)
    top2 = []
    dec.cls.predictions.register_forward_hook(lambda m, i, o: top2.append(torch.topk(o.detach()[:, -1, :], 2, dim=-1).values))
    with torch.no_grad():
        ids, probs = dec(*inp, task_idx=None, sample_mode="greedy")
    t2 = torch.stack(top2, dim=1)                               # [B, T, 2]
    return {"fingerprint": decode_fingerprint(p, inp), "ids": ids.numpy(), "probs": probs.float().numpy(),
            "margin": (t2[..., 0] - t2[..., 1]).float().numpy()}


def run_reference_beam_case(mk, B, T, seed, dk):
    """The reference is pinned to torch 1.1 (Dockerfile:1): integer torch.div floors there (modeling.py:1314 needs that for the
    back pointers) and the forbid mask is moved with .cuda() (:1427) -- both are emulated around the unmodified reference call."""
    p = O.init_params(vocab_size=mk["vocab_size"], layers=mk["layers"], tasks=mk["tasks"], seed=mk["seed"], std=mk["std"])
    dec = ref_loader.build_reference_model(dict(vocab_size=mk["vocab_size"], num_hidden_layers=mk["layers"]), seed=0, decoder=True,
                                           mask_word_id=S.MASK_ID, eos_id=S.SEP_ID, **dk)
    sd = dict(p)
    sd["cls.predictions.decoder.weight"] = p["bert.embeddings.word_embeddings.weight"]
    missing, unexpected = dec.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    dec.eval()
    inp = decode_inputs(B, T, seed)
    true_div, true_cuda = torch.div, torch.Tensor.cuda
    torch.div = lambda a, b, **kw: true_div(a, b, **kw) if (torch.is_tensor(a) and a.is_floating_point()) else true_div(a, b, rounding_mode="floor")
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with torch.no_grad():
            tr = dec(*inp, task_idx=None)
    finally:
        torch.div, torch.Tensor.cuda = true_div, true_cuda
    with torch.no_grad():
        mine = O.beam_search(p, *inp, S.MASK_ID, dk["search_beam_size"], S.SEP_ID, length_penalty=dk["length_penalty"], min_len=dk["min_len"],
                             forbid_duplicate_ngrams=dk["forbid_duplicate_ngrams"], ngram_size=dk["ngram_size"], want_margins=True)
    out = {"fingerprint": decode_fingerprint(p, inp), "margins": np.asarray(mine["margins"], dtype=np.float64),
           "logit_scale": np.asarray(mine["logit_scale"], dtype=np.float64)}
    for k in ("pred_seq", "scores", "wids", "ptrs"):
        out[k] = tr[k].numpy()
    pad = O.pad_traces(mine, inp[3].shape[1])
    assert all(torch.equal(pad[k], tr[k]) for k in ("pred_seq", "wids", "ptrs")), "oracle restatement disagrees with the reference"
    return out


def loader_raw_inputs(seed, Nv=100):
    """Synthetic raw arrays shaped like the dataset's h5 contents: Detectron boxes in pixels (+ a dummy column and a confidence),
    fp16 class probabilities, fp16 fc6 features (README.md:64,118; seq2seq_loader.py:325-330)."""
    rng = np.random.RandomState(seed)
    wh = np.asarray([640.0, 480.0])
    xy1 = rng.uniform(0, 0.7, size=(Nv, 2)) * wh
    xy2 = np.minimum(xy1 + rng.uniform(0.05, 0.3, size=(Nv, 2)) * wh, wh - 1)
    bbox = np.concatenate((xy1, xy2, rng.uniform(0, 1, size=(Nv, 1)), rng.uniform(0.2, 1, size=(Nv, 1))), axis=1).astype(np.float32)
    logits = rng.standard_normal((Nv, 1601)) * 3
    cls = (np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)).astype(np.float16)
    feat = np.abs(rng.standard_normal((Nv, 2048))).astype(np.float16)
    tokens = rng.randint(0, 200, size=64)
    return bbox, cls, feat, tokens


def run_reference_loader_case(mode, n_tokens, seed, max_len_b=20, Nv=100, always_truncate_tail=True, trunc_seg="b", max_pred=3, mask_prob=0.15,
                              mask_image_regions=False, vis_mask_prob=0.25):
    """Runs the UNMODIFIED Preprocess4Seq2seq.__call__ on the synthetic raw arrays, served by the in-memory h5py of ref_loader."""
    import random
    L = ref_loader.load_reference_loader()
    bbox, cls, feat, tokens = loader_raw_inputs(seed, Nv)
    img_id = "COCO_%06d" % seed
    ref_loader.H5_REGISTRY["det_feat" + img_id[-3:] + ".h5"] = {img_id: feat}
    ref_loader.H5_REGISTRY["det_cls" + img_id[-3:] + ".h5"] = {img_id: cls}
    ref_loader.H5_REGISTRY["bbox.h5"] = {img_id: bbox.copy()}
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + ["w%d" % i for i in range(200)]
    idx = {w: i for i, w in enumerate(vocab)}
    proc = L.Preprocess4Seq2seq(max_pred, mask_prob, vocab, lambda toks: [idx[t] for t in toks], max_len=Nv + max_len_b + 3, new_segment_ids=True,
                                truncate_config={"max_len_b": max_len_b, "trunc_seg": trunc_seg, "always_truncate_tail": always_truncate_tail}, mode=mode,
                                len_vis_input=Nv, enable_butd=True, region_bbox_file="bbox.h5", region_det_file_prefix="det",
                                mask_image_regions=mask_image_regions, vis_mask_prob=vis_mask_prob)
    random.seed(seed)
    np.random.seed(seed)            # vis_masked_pos is drawn with np.random.choice (:268)
    out = proc(("/data/" + img_id + ".jpg", ["w%d" % int(t) for t in tokens[:n_tokens]]))
    return out, bbox, cls, feat, min(n_tokens, max_len_b)


def main():
    torch.set_num_threads(8)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    if only:            # python -m oracle.make_golden --only=name[,name]: (re)generate the named forward / backward cases only
        for name in only[0]:
            mk, bk = CASES[name]
            out = run_reference_case(mk, bk)
            path = os.path.join(GOLDEN_DIR, name + ".npz")
            np.savez_compressed(path, **out)
            print("%s: losses=%s  -> %s (%.1f KB)" % (name, out["losses"], path, os.path.getsize(path) / 1024))
        return
    for name, (mode, n_tokens, seed) in LOADER_CASES.items():
        out, bbox, cls, feat, nb = run_reference_loader_case(mode, n_tokens, seed)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, bbox=bbox, cls=cls, vis_pe=out[10].numpy().astype(np.float32), input_mask=out[2].numpy().astype(np.uint8),
                            segment_ids=np.asarray(out[1]), len_b=np.asarray(nb), img_sum=np.asarray(float(out[8].double().sum())))
        print("%s: vis_pe %s mask rows %d -> %s (%.0f KB)" % (name, tuple(out[10].shape), out[2].shape[0], path, os.path.getsize(path) / 1024))
    if "--loader-only" in sys.argv:
        return
    for name, (mk, B, T, seed, dk) in BEAM_CASES.items():
        out = run_reference_beam_case(mk, B, T, seed, dk)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print("%s: pred_seq[0]=%s min margin=%.4f -> %s" % (name, out["pred_seq"][0][:12], out["margins"].min(), path))
    if "--beam-only" in sys.argv:
        return
    for name, (mk, B, T, seed) in DECODE_CASES.items():
        out = run_reference_decode_case(mk, B, T, seed)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print("%s: ids[0]=%s min margin=%.4f -> %s" % (name, out["ids"][0], out["margin"].min(), path))
    if "--decode-only" in sys.argv:
        return
    for name, (mk, bk) in CASES.items():
        out = run_reference_case(mk, bk)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print("%s: losses=%s  -> %s (%.1f KB)" % (name, out["losses"], path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
