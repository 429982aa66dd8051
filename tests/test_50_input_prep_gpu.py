"""GPU parity of the on-device input preparation: vlp_mask_build and vlp_vis_pe_prep against the
loader oracle (pinned to the unmodified reference pipeline) and the reference fixtures, and the model fed with
(RawRegions, MaskSpec) against the same model fed with the dense tensors the reference's DataLoader would have produced."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from oracle import loader_oracle as LO                      # noqa: E402  (checker)
from oracle import vlp_oracle as O                          # noqa: E402
from oracle.make_golden import LOADER_CASES, loader_raw_inputs   # noqa: E402
from vlp_amd import _lib as K                               # noqa: E402
from vlp_amd import synthetic as S                          # noqa: E402
from vlp_amd.input_prep import MaskSpec, RawRegions         # noqa: E402
from vlp_amd.modeling import BertConfig, BertForPreTrainingLossMask   # noqa: E402

DEV = torch.device("cuda:0")
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("L", [123, 167, 64])
def test_mask_build_bit_exact(L):
    Nv = 100 if L > 110 else 20
    rng = np.random.RandomState(L)
    B = 9
    len_b = rng.randint(0, L - Nv - 2, size=B)
    len_b[0], len_b[1] = 0, L - Nv - 3                                  # empty caption; caption filling the whole row
    modes = ["s2s" if rng.rand() < 0.6 else "bi" for _ in range(B)]
    dense = torch.from_numpy(np.stack([LO.attention_mask(Nv, int(nb), L, m) for nb, m in zip(len_b, modes)])).to(DEV)
    spec = MaskSpec.from_lengths(Nv, len_b.tolist(), [m == "s2s" for m in modes], device=DEV)
    Lp = (L + 31) // 32 * 32
    ref, ref_t = torch.empty(B, L, Lp, dtype=torch.uint8, device=DEV), torch.empty(B, Lp, Lp, dtype=torch.uint8, device=DEV)
    K.mask_pack(dense, ref, B, L, Lp, out_t=ref_t)
    out, out_t = torch.full_like(ref, 9), torch.full_like(ref_t, 9)
    K.mask_build(spec.second_st, spec.second_end, spec.is_s2s, out, B, L, Lp, out_t=out_t)
    assert torch.equal(out, ref) and torch.equal(out_t, ref_t)
    assert torch.equal(spec.dense(L), dense)


def test_mask_build_with_masked_regions_bit_exact():
    """--vis_mask_prob > 0: the loader blocks the masked regions' key columns on its dense mask (seq2seq_loader.py:303-304); on the
    device vlp_region_mask_build + vlp_mask_build(region_mask) must produce exactly the packed form of that mask."""
    L, Nv, B, Pm = 167, 100, 7, 25
    rng = np.random.RandomState(5)
    len_b = rng.randint(1, L - Nv - 3, size=B)
    modes = ["s2s" if rng.rand() < 0.6 else "bi" for _ in range(B)]
    vmp = torch.stack([torch.randperm(Nv, generator=torch.Generator().manual_seed(b))[:Pm] + 1 for b in range(B)])
    dense = torch.from_numpy(np.stack([LO.attention_mask(Nv, int(nb), L, m) for nb, m in zip(len_b, modes)]))
    for b in range(B):
        dense[b][:, vmp[b]] = 0
    dense = dense.to(DEV)
    spec = MaskSpec.from_lengths(Nv, len_b.tolist(), [m == "s2s" for m in modes], device=DEV)
    Lp = (L + 31) // 32 * 32
    ref, ref_t = torch.empty(B, L, Lp, dtype=torch.uint8, device=DEV), torch.empty(B, Lp, Lp, dtype=torch.uint8, device=DEV)
    K.mask_pack(dense, ref, B, L, Lp, out_t=ref_t)
    rmask = torch.full((B * Nv,), 9, dtype=torch.uint8, device=DEV)
    K.region_mask_build(vmp.to(DEV), rmask, B, Pm, Nv)
    want = torch.zeros(B, Nv, dtype=torch.uint8)
    want.scatter_(1, vmp - 1, 1)
    assert torch.equal(rmask.cpu().view(B, Nv), want)
    out, out_t = torch.full_like(ref, 9), torch.full_like(ref_t, 9)
    K.mask_build(spec.second_st, spec.second_end, spec.is_s2s, out, B, L, Lp, out_t=out_t, region_mask=rmask, Nv=Nv)
    assert torch.equal(out, ref) and torch.equal(out_t, ref_t)


@pytest.mark.parametrize("name", list(LOADER_CASES.keys()))
@pytest.mark.parametrize("cls_f32", [False, True])
def test_vis_pe_prep_vs_reference_fixture(name, cls_f32):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    B, Nv = 3, 100
    # three images: the fixture's plus two more synthetic ones (oracle-checked), to exercise the per-image normalisation
    raws = [(g["bbox"], g["cls"])] + [loader_raw_inputs(50 + i)[:2] for i in range(B - 1)]
    bbox = torch.from_numpy(np.stack([r[0] for r in raws])).to(DEV)
    cls = torch.from_numpy(np.stack([r[1] for r in raws])).to(DEV)
    if cls_f32:
        cls = cls.float()
    out = torch.full((B * Nv, 1664), 7.0, dtype=torch.float16, device=DEV)
    K.vis_pe_prep(bbox, cls.reshape(B * Nv, 1601), out, B, Nv, 1601, 1664)
    got = out.float().cpu().numpy().reshape(B, Nv, 1664)
    assert np.all(got[:, :, 1607:] == 0)
    for i, (bb, cc) in enumerate(raws):
        want = LO.vis_pe_prepare(bb, cc.astype(np.float32))
        # fp32 arithmetic and ONE fp16 rounding (2^-11 relative; a dominant class is ~sqrt(1601) = 40 after the layer norm)
        assert np.all(np.abs(got[i, :, :1607] - want) <= 5.5e-4 * np.abs(want) + 1e-4)
    ref = g["vis_pe"]                                                 # what the UNMODIFIED reference pipeline returned
    assert np.all(np.abs(got[0, :, :1607] - ref) <= 5.5e-4 * np.abs(ref) + 1e-4)
    # against the reference's own fp16 cast (run_img2txt_dist.py:467): at most one fp16 ulp apart, identical almost everywhere
    ref16 = ref.astype(np.float16).astype(np.float32)
    assert np.all(np.abs(got[0, :, :1607] - ref16) <= 1e-3 * np.abs(ref16) + 1e-4)
    assert np.mean(got[0, :, :1607] == ref16) > 0.98


def test_model_with_raw_inputs_matches_dense_inputs():
    """fwd + bwd of the 2-layer model: (fp16 features, RawRegions, MaskSpec) vs (fp16 features, dense vis_pe, int64 mask)."""
    mk = dict(vocab_size=1024, layers=2, tasks="img2txt", seed=51)
    p = O.init_params(vocab_size=mk["vocab_size"], layers=mk["layers"], tasks=mk["tasks"], seed=mk["seed"])
    B, Nv, max_len_b = 4, 100, 20
    L = Nv + max_len_b + 3
    batch = S.make_batch(batch_size=B, max_len_b=max_len_b, vocab_size=1024, max_pred=3, s2s_prob=0.5, seed=5)
    b = S.batch_to(batch, DEV, half=True)
    # recover (len_b, mode) of the synthetic batch from its dense mask, then rebuild everything from raw arrays
    dense = batch.input_mask.numpy()
    raws = [loader_raw_inputs(70 + i)[:2] for i in range(B)]
    modes, len_b = [], []
    for i in range(B):
        en = int(dense[i, 0].sum())                       # row 0 sees the region block (s2s) or every non-pad column (bi)
        s2s = en == Nv + 2
        modes.append(s2s)
        len_b.append(int(dense[i, :, Nv + 2].sum()) - 1 if s2s else en - Nv - 3)
    spec = MaskSpec.from_lengths(Nv, len_b, modes, device=DEV)
    assert torch.equal(spec.dense(L).cpu(), batch.input_mask)
    vis_pe_dense = torch.from_numpy(np.stack([LO.vis_pe_prepare(bb, cc.astype(np.float32)) for bb, cc in raws])).float()
    raw = RawRegions(torch.from_numpy(np.stack([r[0] for r in raws])).to(DEV), torch.from_numpy(np.stack([r[1] for r in raws])).to(DEV))

    def run(vis_pe, mask):
        cfg = BertConfig(mk["vocab_size"], num_hidden_layers=mk["layers"], type_vocab_size=6, hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0)
        m = BertForPreTrainingLossMask(cfg, enable_butd=True, len_vis_input=Nv, tasks="img2txt", allow_random_fc7=True)
        sd = dict(p)
        sd["cls.predictions.decoder.weight"] = p["bert.embeddings.word_embeddings.weight"]
        m.load_state_dict(sd, strict=True)
        m = m.half().to(DEV).train()
        losses = m(b.img, vis_pe, b.input_ids, b.segment_ids, mask, b.lm_label_ids, b.ans_labels, b.is_next, masked_pos=b.masked_pos,
                   masked_weights=b.masked_weights, task_idx=b.task_idx, vis_masked_pos=b.vis_masked_pos, mask_image_regions=False,
                   drop_worst_ratio=0.0)
        (losses[0] + losses[1] + losses[2]).sum().backward()
        torch.cuda.synchronize()
        return float(losses[0]), {n: q.grad.float().clone() for n, q in m.named_parameters() if q.grad is not None}

    loss_d, grads_d = run(vis_pe_dense.half().to(DEV), b.input_mask)
    loss_r, grads_r = run(raw, spec)
    assert abs(loss_d - loss_r) < 2e-3 * abs(loss_d), (loss_d, loss_r)
    for n in ("vis_pe_embed.0.weight", "bert.encoder.layer.0.attention.self.query.weight", "bert.embeddings.word_embeddings.weight"):
        d = float((grads_d[n] - grads_r[n]).norm() / (grads_d[n].norm() + 1e-12))
        assert d < 2e-2, (n, d)
