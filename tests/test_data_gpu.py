"""GPU checks of the N3 pipeline: BatchPrefetcher delivers exactly the bytes of the packed store / the preprocessor's output, batches are
accepted by the model, and the prefetch thread keeps ahead of the training step."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from vlp_amd import synthetic as S                                # noqa: E402
from vlp_amd.data import BatchPrefetcher, PackedRegionStore, TextPreprocessor, write_packed   # noqa: E402
from vlp_amd.input_prep import MaskSpec, RawRegions                # noqa: E402
from vlp_amd.modeling import BertConfig, BertForPreTrainingLossMask   # noqa: E402
from vlp_amd.optimization_fp16 import FP16_Optimizer_State, FusedAdam   # noqa: E402
from vlp_amd.run_img2txt_dist import train_step                   # noqa: E402

DEV = torch.device("cuda:0")


def make_store(tmp_path, n=24, seed=0):
    rng = np.random.RandomState(seed)
    feats = np.abs(rng.standard_normal((n, 100, 2048))).astype(np.float16)
    cls = rng.dirichlet(np.full(1601, 0.05), size=(n, 100)).astype(np.float16)
    xy1 = rng.uniform(0, 400, size=(n, 100, 2))
    box = np.concatenate((xy1, xy1 + rng.uniform(10, 200, size=(n, 100, 2)), rng.rand(n, 100, 1), rng.uniform(0.2, 1, size=(n, 100, 1))), axis=2)
    ids = ["img%05d" % i for i in range(n)]
    write_packed(str(tmp_path), ids, feats, cls, box.astype(np.float32))
    examples = [(ids[i % n], rng.randint(1000, 2000, size=rng.randint(3, 30)).tolist()) for i in range(3 * n)]
    return PackedRegionStore(str(tmp_path)), examples, feats, cls, box.astype(np.float32), ids


def procs(vocab=2048, max_len_b=20):
    kw = dict(max_pred=3, mask_prob=0.15, vocab_size=vocab, cls_id=S.CLS_ID, sep_id=S.SEP_ID, mask_id=S.MASK_ID, unk_id=S.UNK_ID,
              max_len=100 + max_len_b + 3, max_len_b=max_len_b, len_vis_input=100)
    return TextPreprocessor(mode="s2s", **kw), TextPreprocessor(mode="bi", **kw)


def test_prefetcher_delivers_exact_batches(tmp_path):
    store, examples, feats, cls, box, ids = make_store(tmp_path)
    p_s2s, p_bi = procs()
    B, steps = 4, 5
    random.seed(11)
    pf = BatchPrefetcher(store, examples, B, p_s2s, p_bi, s2s_prob=0.5, device=DEV, steps=steps, seed=3)
    got = []
    for batch in pf:
        torch.cuda.synchronize()
        got.append([t.cpu() if torch.is_tensor(t) else type(t)(*(x.cpu() for x in t)) for t in batch])
    assert len(got) == steps
    # replay the same sample stream synchronously
    order = list(range(len(examples)))
    random.Random(3).shuffle(order)
    random.seed(11)
    row = {k: i for i, k in enumerate(ids)}
    for s in range(steps):
        for j in range(B):
            img_id, toks = examples[order[(s * B + j) % len(order)]]
            proc = random.choices([p_s2s, p_bi], weights=[0.5, 0.5])[0]
            t = proc(toks)
            g = got[s]
            assert g[0][j].tolist() == t["input_ids"] and g[1][j].tolist() == t["segment_ids"]
            assert g[3][j].tolist() == t["masked_ids"] and g[4][j].tolist() == t["masked_pos"] and g[5][j].tolist() == t["masked_weights"]
            assert int(g[2].second_st[j]) == t["len_a"] + 2 and int(g[2].second_end[j]) == t["len_a"] + t["len_b"] + 3
            assert int(g[2].is_s2s[j]) == int(t["is_s2s"]) and int(g[7][j]) == t["task_idx"] and int(g[6][j]) == -1
            r = row[img_id]
            assert np.array_equal(g[8][j].numpy(), feats[r]) and np.array_equal(g[10].cls_prob[j].numpy(), cls[r])
            assert np.array_equal(g[10].bbox[j].numpy(), box[r])


def test_training_from_the_prefetcher(tmp_path):
    store, examples, *_ = make_store(tmp_path, n=16, seed=1)
    p_s2s, p_bi = procs()
    cfg = BertConfig(2048, num_hidden_layers=2, type_vocab_size=6)
    model = BertForPreTrainingLossMask(cfg, enable_butd=True, len_vis_input=100, tasks="img2txt", allow_random_fc7=True).half().to(DEV).train()
    named = list(model.named_parameters())
    nd = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    groups = [{"params": [p for n, p in named if not any(x in n for x in nd)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(x in n for x in nd)], "weight_decay": 0.0}]
    opt = FP16_Optimizer_State(FusedAdam(groups, lr=3e-4, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True)
    random.seed(5)
    losses = []
    for batch in BatchPrefetcher(store, examples, 8, p_s2s, p_bi, s2s_prob=0.75, device=DEV, steps=12, seed=1):
        assert isinstance(batch[2], MaskSpec) and isinstance(batch[10], RawRegions) and batch[8].dtype == torch.float16
        lt = train_step(model, opt, batch, 3e-4)
        losses.append(float(lt[0].detach()))
    assert all(l == l for l in losses) and sum(losses[-3:]) < sum(losses[:3]), losses


def test_entry_script_from_packed_features(tmp_path, monkeypatch):
    """python -m vlp_amd.run_img2txt_dist --packed_features ... --token_file ...: one epoch over a small packed store.  Like the
    reference (modeling.py:1008-1014) a real-data run insists on detectron_weights/fc7_{w,b}.pkl; the test opts out explicitly."""
    monkeypatch.setenv("VLP_ALLOW_RANDOM_FC7", "1")
    import json
    import os
    from vlp_amd import run_img2txt_dist as R
    store_dir = os.path.join(tmp_path, "store")
    os.makedirs(store_dir)
    _, examples, *_ = make_store(store_dir, n=12, seed=2)
    tok = os.path.join(tmp_path, "tokens.json")
    json.dump([[i, t] for i, t in examples], open(tok, "w"))
    out = os.path.join(tmp_path, "run")
    R.main(["--output_dir", out, "--fp16", "--enable_butd", "--new_segment_ids", "--from_scratch", "--max_len_b", "20", "--train_batch_size", "4",
            "--num_train_epochs", "1", "--num_hidden_layers", "2", "--len_vis_input", "100", "--packed_features", store_dir, "--token_file", tok,
            "--s2s_prob", "0.75", "--bi_prob", "0.25", "--always_truncate_tail"])
    assert os.path.exists(os.path.join(out, "model.1.bin")) and os.path.exists(os.path.join(out, "optim.1.bin"))
