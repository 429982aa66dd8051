"""GPU parity of the incremental greedy decoder (modeling.py:1189-1253):
the K/V-cache helper kernels one by one, then vlp_amd.modeling.BertForSeq2SeqDecoder end to end against
(a) fixtures the UNMODIFIED reference decoder produced (tests/golden/decode_*.npz) and
(b) the oracle restatement (hidden-state history caches, as the reference) run on the same device.

Token ids are integer results: they must be identical wherever the reference's own top-1/top-2 logit margin
exceeds the fp16 evaluation noise (MARGIN_TOL); a flip is only tolerated at a step whose margin is below it,
and the sample is then compared only up to that step (later inputs differ).  The returned scores (max logits)
must agree to SCORE_TOL relative.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from oracle import vlp_oracle as O                         # noqa: E402  (checker)
from oracle.make_golden import DECODE_CASES, decode_inputs, decode_fingerprint   # noqa: E402  (pure helpers)
from vlp_amd import _lib as K                              # noqa: E402
from vlp_amd import synthetic as S                         # noqa: E402
from vlp_amd.modeling import BertConfig, BertForSeq2SeqDecoder   # noqa: E402

DEV = torch.device("cuda:0")
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MARGIN_TOL = 1.5e-2      # logits ~4: fp16 rounding of logits alone is 2e-3..4e-3, plus accumulated activation error
SCORE_TOL = 4e-3         # relative, on the max logit


# ------------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Lq,Lk,st", [(3, 103, 103, 0), (4, 2, 117, 115), (2, 2, 32, 30), (1, 5, 33, 28)])
def test_mask_pack_rect(B, Lq, Lk, st):
    Lt = 140
    g = torch.Generator().manual_seed(Lk)
    full = (torch.rand(B, Lt, Lt, generator=g) < 0.6).long().to(DEV)
    view = full[:, st:st + Lq, :Lk]
    Lkp = (Lk + 31) // 32 * 32
    out = torch.full((B, Lq, Lkp), 9, dtype=torch.uint8, device=DEV)
    K.mask_pack_rect(view, out, B, Lq, Lk, Lkp)
    exp = torch.full((B, Lq, Lkp), 2, dtype=torch.uint8, device=DEV)
    exp[:, :, :Lk] = (view != 0).to(torch.uint8)
    assert torch.equal(out, exp)


def test_kv_append():
    B, T, H, Lcap, st = 3, 5, 128, 40, 17
    qkv = torch.randn(B * T, 3 * H, device=DEV).half()
    cache = torch.zeros(B, Lcap, 2 * H, device=DEV, dtype=torch.float16)
    K.kv_append(qkv, 3 * H, cache, Lcap, B, T, st, H)
    exp = torch.zeros_like(cache)
    exp[:, st:st + T] = qkv.view(B, T, 3 * H)[:, :, H:]
    assert torch.equal(cache, exp)
    with pytest.raises(RuntimeError):
        K.kv_append(qkv, 3 * H, cache, Lcap, B, T, Lcap - 2, H)         # would run past the cache


@pytest.mark.parametrize("rows,V", [(7, 1000), (64, 28996), (1, 1)])
def test_argmax_rows(rows, V):
    Vp = (V + 63) // 64 * 64
    g = torch.Generator().manual_seed(V)
    logits = torch.randn(rows, Vp, generator=g).half().to(DEV)
    logits[:, V:] = 100.0                      # padding columns must be ignored
    if V > 10:
        logits[0, 5] = logits[0, 9] = 50.0     # tie: first maximum wins (torch.max semantics on CPU / argmax)
    ids = torch.zeros(rows, 3, dtype=torch.long, device=DEV)
    vals = torch.zeros(rows, 2, dtype=torch.float32, device=DEV)
    K.argmax_rows(logits, Vp, rows, V, ids[:, 1], vals[:, 0])
    ev, ei = torch.max(logits[:, :V].float(), dim=-1)
    assert torch.equal(vals[:, 0], ev)
    assert torch.equal(logits[torch.arange(rows), ids[:, 1]].float(), ev)
    if V > 10:
        assert int(ids[0, 1]) == 5
    assert int(ids[:, 0].abs().sum()) == 0 and int(ids[:, 2].abs().sum()) == 0


@pytest.mark.parametrize("B,Lq,Lk,Lcap,heads", [(2, 103, 103, 122, 12), (3, 2, 110, 122, 12), (2, 2, 33, 64, 2), (1, 64, 200, 256, 4)])
def test_attn_decode_vs_torch(B, Lq, Lk, Lcap, heads):
    H = heads * 64
    g = torch.Generator().manual_seed(Lk * 7 + Lq)
    q = (torch.randn(B * Lq, 3 * H, generator=g) * 0.8).half().to(DEV)          # Q lives in a packed qkv buffer
    cache = (torch.randn(B, Lcap, 2 * H, generator=g) * 0.8).half().to(DEV)
    mask = (torch.rand(B, Lq, Lk, generator=g) < 0.7).long().to(DEV)
    mask[:, :, 0] = 1
    Lkp = (Lk + 31) // 32 * 32
    mb = torch.empty(B, Lq, Lkp, dtype=torch.uint8, device=DEV)
    K.mask_pack_rect(mask, mb, B, Lq, Lk, Lkp)
    ctx = torch.zeros(B * Lq, H, dtype=torch.float16, device=DEV)
    K.attn_decode(q, 3 * H, Lq, cache, cache[:, :, H:], 2 * H, Lcap, mb, ctx, B, Lq, Lk, heads, 0.125)
    qf = q.view(B, Lq, 3 * H)[:, :, :H].float().view(B, Lq, heads, 64).transpose(1, 2)
    kf = cache[:, :Lk, :H].float().view(B, Lk, heads, 64).transpose(1, 2)
    vf = cache[:, :Lk, H:].float().view(B, Lk, heads, 64).transpose(1, 2)
    sc = qf @ kf.transpose(-1, -2) * 0.125 + (1.0 - mask[:, None].float()) * -10000.0
    ref = (torch.softmax(sc, dim=-1) @ vf).transpose(1, 2).reshape(B * Lq, H)
    err = float((ctx.float() - ref).abs().max())
    assert err < 4e-3, err


def test_embed_position_ids():
    B, T, H, V = 3, 2, 128, 50
    g = torch.Generator().manual_seed(3)
    word, pos, typ = [torch.randn(n, H, generator=g).half().to(DEV) for n in (V, 64, 6)]
    ids = torch.randint(0, V, (B, T), generator=g).to(DEV)
    seg = torch.randint(0, 6, (B, T), generator=g).to(DEV)
    pid = torch.tensor([[40, 41], [7, 8], [62, 63]], device=DEV)
    out = torch.zeros(B * T, H, dtype=torch.float16, device=DEV)
    K.embed_fwd(ids, seg, word, pos, typ, None, None, out, B, T, 0, H, position_ids=pid)
    exp = (word[ids].float() + pos[pid].float() + typ[seg].float()).half().view(B * T, H)
    assert torch.equal(out, exp)


# ------------------------------------------------------------------------------------------------
# end to end
# ------------------------------------------------------------------------------------------------
def build_decoder(p, mk, Nv=100):
    cfg = BertConfig(mk["vocab_size"], num_hidden_layers=mk["layers"], type_vocab_size=6, hidden_dropout_prob=0.1,
                     attention_probs_dropout_prob=0.1)
    m = BertForSeq2SeqDecoder(cfg, mask_word_id=S.MASK_ID, eos_id=S.SEP_ID, enable_butd=True, len_vis_input=Nv)
    sd = dict(p)
    sd["cls.predictions.decoder.weight"] = p["bert.embeddings.word_embeddings.weight"]
    m.load_state_dict(sd, strict=True)
    return m.half().to(DEV).eval()


def compare_decodes(ids, vals, ref_ids, ref_vals, margin):
    """ids / ref_ids [B, n]; margin [B, n] = reference top-1 minus top-2 logit.  Returns (#tokens compared, #tolerated flips)."""
    ids, vals = np.asarray(ids), np.asarray(vals, dtype=np.float64)
    ref_ids, ref_vals, margin = np.asarray(ref_ids), np.asarray(ref_vals, dtype=np.float64), np.asarray(margin, dtype=np.float64)
    compared, flips = 0, 0
    for b in range(ids.shape[0]):
        for s in range(ids.shape[1]):
            rel = abs(vals[b, s] - ref_vals[b, s]) / (abs(ref_vals[b, s]) + 1e-30)
            assert rel < SCORE_TOL, "sample %d step %d: score %g vs %g" % (b, s, vals[b, s], ref_vals[b, s])
            compared += 1
            if ids[b, s] != ref_ids[b, s]:
                assert margin[b, s] < MARGIN_TOL, "sample %d step %d: token %d vs %d although the reference margin is %g" % (
                    b, s, ids[b, s], ref_ids[b, s], margin[b, s])
                flips += 1
                break                               # the continuation was fed a different token
    return compared, flips


@pytest.mark.parametrize("name", list(DECODE_CASES.keys()))
def test_greedy_decode_vs_reference_fixture(name):
    mk, B, T, seed = DECODE_CASES[name]
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    p = O.init_params(vocab_size=mk["vocab_size"], layers=mk["layers"], tasks=mk["tasks"], seed=mk["seed"], std=mk["std"])
    inp = decode_inputs(B, T, seed)
    if not np.allclose(decode_fingerprint(p, inp), g["fingerprint"], rtol=1e-9, atol=0):
        pytest.skip("RNG stream differs from the one that generated the fixture")
    m = build_decoder(p, mk)
    img, vis_pe, input_ids, token_type, pos, am = [t.to(DEV) for t in inp]
    ids, vals = m(img.half(), vis_pe.half(), input_ids, token_type, pos, am, task_idx=None, sample_mode="greedy")
    torch.cuda.synchronize()
    assert ids.shape == (B, T) and ids.dtype == torch.long and vals.shape == (B, T)
    compared, flips = compare_decodes(ids.cpu(), vals.cpu(), g["ids"], g["probs"], g["margin"])
    assert compared >= 0.8 * B * T, (compared, flips)
    # a second call reuses workspaces / caches and must reproduce itself bit for bit
    ids2, vals2 = m(img.half(), vis_pe.half(), input_ids, token_type, pos, am)
    assert torch.equal(ids, ids2) and torch.equal(vals, vals2)


def test_greedy_decode_vs_oracle_ragged_mask():
    """Batch 5 with a per-sample number of valid regions (attention mask hides the rest, as Preprocess4Seq2seqDecoder does for
    images with fewer boxes) and non-trivial position ids: HIP decode vs the oracle with hidden-state history in fp32."""
    mk = dict(vocab_size=1536, layers=3, tasks="img2txt", seed=31, std=0.05)
    p = O.init_params(vocab_size=mk["vocab_size"], layers=mk["layers"], tasks=mk["tasks"], seed=mk["seed"], std=mk["std"])
    B, T, Nv = 5, 9, 100
    img, vis_pe, input_ids, token_type, pos, am = decode_inputs(B, T, 77)
    valid = [100, 37, 64, 99, 1]
    for b, n in enumerate(valid):
        am[b, :, 1 + n:1 + Nv] = 0
    pos = pos + 3
    m = build_decoder(p, mk)
    dv = [t.to(DEV) for t in (img, vis_pe, input_ids, token_type, pos, am)]
    ids, vals = m(dv[0].half(), dv[1].half(), *dv[2:])
    pd = {k: v.to(DEV).half().float() for k, v in p.items()}          # the fp16-rounded weights, evaluated in fp32
    top2 = []
    orig = O.lm_head

    def spy(pp, x):
        out = orig(pp, x)
        top2.append(torch.topk(out[:, -1, :], 2, dim=-1).values)
        return out
    O.lm_head = spy
    try:
        with torch.no_grad():
            oids, ovals = O.greedy_decode(pd, dv[0].half().float(), dv[1].half().float(), *dv[2:], S.MASK_ID, len_vis_input=Nv)
    finally:
        O.lm_head = orig
    t2 = torch.stack(top2, dim=1)
    compared, flips = compare_decodes(ids.cpu(), vals.cpu(), oids.cpu(), ovals.cpu(), (t2[..., 0] - t2[..., 1]).cpu())
    assert compared >= 0.8 * B * T, (compared, flips)


def test_decoder_rejects_unbuilt_modes():
    cfg = BertConfig(512, num_hidden_layers=1, type_vocab_size=6)
    m = BertForSeq2SeqDecoder(cfg, mask_word_id=S.MASK_ID, enable_butd=True, len_vis_input=100, search_beam_size=2).half().to(DEV)
    inp = [t.to(DEV) for t in decode_inputs(1, 2, 1)]
    with pytest.raises(NotImplementedError):
        m(*inp)
    m.search_beam_size = 1
    with pytest.raises(NotImplementedError):
        m(*inp, sample_mode="sample")
