"""GPU parity of the incremental greedy decoder (modeling.py:1189-1253):
the K/V-cache helper kernels one by one, then vlp_amd.modeling.BertForSeq2SeqDecoder end to end against
(a) fixtures the UNMODIFIED reference decoder produced (tests/golden/decode_*.npz) and
(b) the oracle restatement (hidden-state history caches, as the reference) run on the same device.

Token ids are integer results: they must be identical wherever the reference's own top-1/top-2 logit margin
exceeds the fp16 evaluation noise (MARGIN_TOL); a flip is only tolerated at a step whose margin is below it,
and the sample is then compared only up to that step (later inputs differ).  The returned scores (max logits)
must agree to SCORE_TOL relative.
"""
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


# ---- round 6: the burst kernels of a token step (csrc/decode.hip) against plain torch fp32 ----------------------------------------------
def _rand16(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().to(DEV)


@pytest.mark.parametrize("M,N,Kd,act", [(128, 2304, 768, 0), (128, 3072, 768, 2), (70, 96, 128, 0), (384, 768, 768, 2), (64, 1000, 192, 0), (1, 32, 64, 0), (70, 100, 128, 0), (64, 28996, 768, 0)])
def test_dec_gemm_plain_and_gelu(M, N, Kd, act):
    """Y = act(X W^T + b): fp32 torch reference on the same fp16 operands; ragged M (not a multiple of 64) and N (not a multiple of 8 / 32);
    rows / columns outside the problem are not written."""
    x, w, b = _rand16(M, Kd, seed=1), _rand16(N, Kd, scale=0.05, seed=2), _rand16(N, seed=3)
    ldy = (N + 7) // 8 * 8 + 8
    y = torch.full((M + 3, ldy), 7.0, device=DEV, dtype=torch.float16)
    K.dec_gemm(x, w, M, N, Kd, y=y, bias=b, act=K.ACT_GELU if act else K.ACT_NONE)
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t() + b.float()
    if act:
        ref = torch.nn.functional.gelu(ref)
    got = y[:M, :N].float()
    assert float((got - ref).abs().max()) <= 2e-3 * max(1.0, float(ref.abs().max())) + 2e-3
    n8 = (N + 7) // 8 * 8
    assert torch.all(y[M:] == 7.0) and torch.all(y[:M, n8:] == 7.0)                 # nothing outside [M, roundup8(N)] is touched
    assert torch.all(y[:M, N:n8] == 0)                                              # the pad columns of the last vector are written as zero


def test_dec_gemm_qkv_writes_the_kv_cache():
    """The QKV form: columns < H land in Y, columns >= H in the K | V cache row of (sequence, absolute position) -- what a plain GEMM
    followed by vlp_kv_append produces, bit for bit."""
    R, T, H, Lcap, st = 6, 2, 256, 40, 17
    M = R * T
    x, w, b = _rand16(M, H, seed=4), _rand16(3 * H, H, scale=0.05, seed=5), _rand16(3 * H, seed=6)
    y = torch.zeros(M, 3 * H, device=DEV, dtype=torch.float16)
    cache = torch.full((R, Lcap, 2 * H), 3.0, device=DEV, dtype=torch.float16)
    K.dec_gemm(x, w, M, 3 * H, H, y=y, bias=b, kv_cache=cache, kv_col0=H, kv_Lcap=Lcap, kv_T=T, kv_start=st)
    full = torch.zeros(M, 3 * H, device=DEV, dtype=torch.float16)
    K.dec_gemm(x, w, M, 3 * H, H, y=full, bias=b)
    cache2 = torch.full((R, Lcap, 2 * H), 3.0, device=DEV, dtype=torch.float16)
    K.kv_append(full, 3 * H, cache2, Lcap, R, T, st, H)
    torch.cuda.synchronize()
    assert torch.equal(y[:, :H], full[:, :H]) and torch.all(y[:, H:] == 0)
    assert torch.equal(cache, cache2)
    ref = (x.float() @ w.float().t() + b.float())
    assert float((full.float() - ref).abs().max()) <= 3e-3 * float(ref.abs().max())


@pytest.mark.parametrize("M,H,Kd,S", [(128, 768, 3072, 4), (128, 768, 768, 4), (37, 256, 512, 2), (192, 512, 128, 1)])
def test_dec_split_gemm_plus_reduce_layernorm(M, H, Kd, S):
    """Split form + vlp_dec_reduce_ln = LayerNorm(fp16(X W^T + bias + residual)): against torch in fp32 with the same fp16 rounding of the
    pre-LayerNorm sum; deterministic (two runs are bit-identical)."""
    x, w = _rand16(M, Kd, seed=7), _rand16(H, Kd, scale=0.03, seed=8)
    b, res, ga, be = _rand16(H, seed=9), _rand16(M, H, seed=10), (1 + 0.1 * _rand16(H, seed=11).float()).half(), _rand16(H, scale=0.1, seed=12)
    outs = []
    for _ in range(2):
        slab = torch.full((S, M, H), float("nan"), device=DEV, dtype=torch.float32)
        y = torch.empty(M, H, device=DEV, dtype=torch.float16)
        K.dec_gemm(x, w, M, H, Kd, slab=slab, splits=S)
        K.dec_reduce_ln(slab, S, b, res, ga, be, y, M, H, eps=1e-5)
        torch.cuda.synchronize()
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    pre = (x.float() @ w.float().t() + b.float() + res.float()).half().float()
    ref = torch.nn.functional.layer_norm(pre, (H,), ga.float(), be.float(), 1e-5)
    # a pre-LayerNorm element that sits on an fp16 rounding boundary may round the other way under another summation order: compare with
    # the LayerNorm of OUR fp16 pre-activation's neighbourhood -> tolerance of one fp16 ulp of the pre-activation, amplified by rstd
    err = float((outs[0].float() - ref).abs().max())
    assert err <= 6e-3, err
    assert float((outs[0].float() - ref).abs().mean()) <= 4e-4


@pytest.mark.parametrize("M,N,Kd,act", [(128, 3072, 768, 2), (70, 96, 256, 0), (192, 768, 512, 0), (5, 40, 64, 2)])
def test_dec_gemm_layernorm_prologue_and_residual(M, N, Kd, act):
    """(a) residual epilogue: Y = fp16(X W^T + b + R) in one rounding; (b) LayerNorm prologue: Y = act(LayerNorm(P) W^T + b) with the normalised rows also
    written to ln_out -- against torch fp32 on the fp16 operands, and against the two-launch form (layernorm_fwd, then the plain kernel) which
    rounds at the same points."""
    x, w, b = _rand16(M, Kd, seed=21), _rand16(N, Kd, scale=0.05, seed=22), _rand16(N, seed=23)
    res = _rand16(M, N, seed=24)
    y = torch.empty(M, N, device=DEV, dtype=torch.float16)
    K.dec_gemm(x, w, M, N, Kd, y=y, bias=b, residual=res)
    ref = x.float() @ w.float().t() + b.float() + res.float()
    assert float((y.float() - ref).abs().max()) <= 3e-3 * max(1.0, float(ref.abs().max()))
    # prologue
    pre = _rand16(M, Kd, scale=2.0, seed=25) + 0.5
    ga, be = (1 + 0.1 * _rand16(Kd, seed=26).float()).half(), _rand16(Kd, scale=0.1, seed=27)
    y1 = torch.empty(M, N, device=DEV, dtype=torch.float16)
    xn = torch.full((M + 2, Kd), 9.0, device=DEV, dtype=torch.float16)
    K.dec_gemm(pre, w, M, N, Kd, y=y1, bias=b, act=K.ACT_GELU if act else K.ACT_NONE, ln_gamma=ga, ln_beta=be, ln_eps=1e-5, ln_out=xn)
    xr = torch.empty(M, Kd, device=DEV, dtype=torch.float16)
    if Kd % 256 == 0:
        K.layernorm_fwd(pre, ga, be, xr, M, Kd)
    else:
        xr = torch.nn.functional.layer_norm(pre.float(), (Kd,), ga.float(), be.float(), 1e-5).half()
    y2 = torch.empty(M, N, device=DEV, dtype=torch.float16)
    K.dec_gemm(xr, w, M, N, Kd, y=y2, bias=b, act=K.ACT_GELU if act else K.ACT_NONE)
    torch.cuda.synchronize()
    refn = torch.nn.functional.layer_norm(pre.float(), (Kd,), ga.float(), be.float(), 1e-5)
    assert float((xn[:M].float() - refn).abs().max()) <= 4e-3 and torch.all(xn[M:] == 9.0)
    assert float((xn[:M].float() - xr.float()).abs().max()) <= 2e-3                  # an fp16 ulp at most where the statistics' summation order flips a rounding
    assert float((y1.float() - y2.float()).abs().max()) <= 4e-3 * max(1.0, float(y2.float().abs().max()))


@pytest.mark.parametrize("rows,V", [(64, 28996), (3, 1000), (5, 7), (2, 29056)])
def test_argmax_rows2(rows, V):
    g = torch.Generator().manual_seed(rows * 131 + V)
    ld = (V + 63) // 64 * 64
    x = torch.randn(rows, ld, generator=g).half()
    x[0, V // 3] = 9.0
    if V > 10:
        x[0, V // 3 + 5] = 9.0                                   # a tie: the FIRST maximum wins (torch.argmax on CPU, modeling.py:1228)
    x[:, V:] = 100.0                                             # pad columns must be ignored
    xd = x.to(DEV)
    a, b_, v = torch.zeros(rows, 3, dtype=torch.long, device=DEV), torch.zeros(rows, 2, dtype=torch.long, device=DEV), torch.zeros(rows, device=DEV)
    K.argmax_rows2(xd, ld, rows, V, a[:, 1], b_[:, 0], v)
    torch.cuda.synchronize()
    ref_v, ref_i = x[:, :V].float().max(dim=1)
    first = torch.tensor([int((x[r, :V].float() == ref_v[r]).nonzero()[0]) for r in range(rows)])
    assert torch.equal(a[:, 1].cpu(), first) and torch.equal(b_[:, 0].cpu(), first) and torch.equal(v.cpu(), ref_v)
    assert torch.all(a[:, 0] == 0) and torch.all(a[:, 2] == 0) and torch.all(b_[:, 1] == 0)


def test_greedy_decode_fused_token_steps_equal_the_unfused_path():
    """The same decoder with the token steps on the round-6 burst kernels and on the round-2 split-K path: identical token ids wherever the
    top-1 / top-2 margin exceeds the fp16 noise, scores to SCORE_TOL (the two differ in fp32 summation order only)."""
    from vlp_amd.engine import Engine
    mk = dict(vocab_size=1536, layers=3, tasks="img2txt", seed=31, std=0.05)
    p = O.init_params(vocab_size=mk["vocab_size"], layers=mk["layers"], tasks=mk["tasks"], seed=mk["seed"], std=mk["std"])
    inp = [t.to(DEV) for t in decode_inputs(6, 8, 55)]
    outs = {}
    old = Engine.DECODE_FUSED
    try:
        for fused in (True, False):
            Engine.DECODE_FUSED = fused
            m = build_decoder(p, mk)
            for _ in range(3):                                  # plain call, graph capture, replay
                ids, vals = m(inp[0].half(), inp[1].half(), *inp[2:])
            torch.cuda.synchronize()
            outs[fused] = (ids.cpu(), vals.cpu())
    finally:
        Engine.DECODE_FUSED = old
    ids_f, vals_f = outs[True]
    ids_u, vals_u = outs[False]
    same = ids_f == ids_u
    assert float(same.float().mean()) >= 0.95, same
    first_diff = [(int((~same[b]).nonzero()[0]) if (~same[b]).any() else ids_f.shape[1]) for b in range(ids_f.shape[0])]
    for b, n in enumerate(first_diff):                           # up to the first differing token the inputs were identical
        assert torch.allclose(vals_f[b, :n], vals_u[b, :n], rtol=SCORE_TOL, atol=SCORE_TOL)


# (Lq <= 2, Lk <= 128: the wave-per-(sequence, head) token-step kernel of round 6 -- every group count 1 .. 16 of its 8-row groups is hit below)
@pytest.mark.parametrize("B,Lq,Lk,Lcap,heads", [(2, 103, 103, 122, 12), (3, 2, 110, 122, 12), (2, 2, 33, 64, 2), (1, 64, 200, 256, 4)] +
                         [(2, 1 + (lk % 2), lk, 128, 3) for lk in (1, 7, 8, 9, 16, 23, 31, 40, 47, 50, 63, 64, 65, 72, 81, 95, 104, 111, 113, 120, 127, 128)])
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


def test_attn_decode_shared_prefix():
    """Beam form: rows < n_prefix from the per-sample cache, the rest from the per-beam cache == everything from one expanded cache."""
    B, Kb, Lq, Lk, Lcap, heads, npre = 2, 3, 2, 110, 122, 12, 102
    H, R = heads * 64, 2 * 3
    g = torch.Generator().manual_seed(9)
    q = (torch.randn(R * Lq, 3 * H, generator=g) * 0.8).half().to(DEV)
    sample_cache = (torch.randn(B, Lcap, 2 * H, generator=g) * 0.8).half().to(DEV)
    beam_cache = (torch.randn(R, Lcap, 2 * H, generator=g) * 0.8).half().to(DEV)      # its rows < npre are garbage on purpose
    full = beam_cache.clone()
    full[:, :npre] = sample_cache.repeat_interleave(Kb, dim=0)[:, :npre]
    mask = (torch.rand(R, Lq, Lk, generator=g) < 0.8).long().to(DEV)
    mask[:, :, 0] = 1
    Lkp = (Lk + 31) // 32 * 32
    mb = torch.empty(R, Lq, Lkp, dtype=torch.uint8, device=DEV)
    K.mask_pack_rect(mask, mb, R, Lq, Lk, Lkp)
    ref, got = torch.zeros(R * Lq, H, dtype=torch.float16, device=DEV), torch.zeros(R * Lq, H, dtype=torch.float16, device=DEV)
    K.attn_decode(q, 3 * H, Lq, full, full[:, :, H:], 2 * H, Lcap, mb, ref, R, Lq, Lk, heads, 0.125)
    K.attn_decode(q, 3 * H, Lq, beam_cache, beam_cache[:, :, H:], 2 * H, Lcap, mb, got, R, Lq, Lk, heads, 0.125, k_prefix=sample_cache,
                  v_prefix=sample_cache[:, :, H:], prefix_rows=Lcap, n_prefix=npre, beams=Kb)
    assert torch.equal(ref, got)


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
    # call 2 records the launch plans of the token steps, call 3 and 4 replay them (with different caller tensors holding the same values)
    for rep in range(3):
        ids2, vals2 = m(img.half().clone(), vis_pe.half().clone(), input_ids.clone(), token_type.clone(), pos.clone(), am.clone())
        assert torch.equal(ids, ids2) and torch.equal(vals, vals2), rep
    # replayed plans must follow NEW inputs: another image batch gives the oracle-consistent answer of that batch, not a stale one
    img_b = torch.flip(img, dims=[0]).half()
    ids_b, _ = m(img_b, torch.flip(vis_pe, dims=[0]).half(), input_ids, token_type, pos, am)
    assert torch.equal(ids_b, torch.flip(ids, dims=[0]))


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


def test_decoder_rejects_unknown_mode_and_cpu_inputs():
    cfg = BertConfig(512, num_hidden_layers=1, type_vocab_size=6)
    m = BertForSeq2SeqDecoder(cfg, mask_word_id=S.MASK_ID, enable_butd=True, len_vis_input=100).half().to(DEV)
    inp = decode_inputs(1, 2, 1)
    with pytest.raises(NotImplementedError):
        m(*[t.to(DEV) for t in inp], sample_mode="nucleus")
    with pytest.raises(RuntimeError):
        m(*inp)                                              # CPU tensors: there is no CPU path


def test_sample_rows_distribution_and_logprob():
    rows, V = 4096, 11
    Vp = 64
    base = torch.tensor([2.0, 1.0, 0.0, -1.0, 0.5, 3.0, -2.0, 0.0, 1.5, -0.5, 2.5])
    logits = torch.zeros(rows, Vp).half()
    logits[:, :V] = base.half()
    logits[:, V:] = 50.0
    logits = logits.to(DEV)
    ids = torch.zeros(rows, dtype=torch.long, device=DEV)
    lp = torch.zeros(rows, dtype=torch.float32, device=DEV)
    K.sample_rows(logits, Vp, rows, V, 1234, 7001, ids, lp)
    ref_lp = torch.log_softmax(base.half().float(), dim=-1).to(DEV)
    assert int(ids.max()) < V
    assert float((lp - ref_lp[ids]).abs().max()) < 1e-5
    freq = torch.bincount(ids, minlength=V).float() / rows
    prob = ref_lp.exp()
    sigma = torch.sqrt(prob * (1 - prob) / rows)
    assert float(((freq - prob).abs() / sigma).max()) < 5.0, (freq, prob)          # every bin within 5 sigma
    ids2 = torch.zeros_like(ids)
    K.sample_rows(logits, Vp, rows, V, 1234, 7001, ids2, lp)
    assert torch.equal(ids, ids2)                                                  # pure function of (seed, stream, row, column)
    K.sample_rows(logits, Vp, rows, V, 1235, 7001, ids2, lp)
    assert not torch.equal(ids, ids2)


def test_sample_mode_end_to_end():
    """sample_mode='sample' (:1229-1235): ids are draws (not comparable to torch.multinomial's stream); the returned values must be
    the log-probabilities of the drawn ids under the model, checked by teacher-forcing the drawn sequence through the oracle."""
    mk = dict(vocab_size=1024, layers=2, tasks="img2txt", seed=41, std=0.05)
    p = O.init_params(vocab_size=mk["vocab_size"], layers=mk["layers"], tasks=mk["tasks"], seed=mk["seed"], std=mk["std"])
    B, T = 3, 6
    inp = decode_inputs(B, T, 99)
    m = build_decoder(p, mk)
    dv = [t.to(DEV) for t in inp]
    ids, lps = m(dv[0].half(), dv[1].half(), *dv[2:], sample_mode="sample")
    assert ids.shape == (B, T) and lps.shape == (B, T) and float(lps.max()) <= 0.0
    ids_b, _ = m(dv[0].half(), dv[1].half(), *dv[2:], sample_mode="sample")
    assert not torch.equal(ids, ids_b)                      # a new draw every call
    # teacher-forced check: full (non-incremental) oracle forward over [prefix, drawn tokens with a [MASK] at the position to predict]
    pd = {k: v.to(DEV).half().float() for k, v in p.items()}
    in_len = dv[2].shape[1]
    vf, vp = O.vis_embed(pd, dv[0].half().float()), O.vis_pe_embed(pd, dv[1].half().float())
    for s in range(T):
        x = torch.cat((dv[2], ids[:, :s], torch.full((B, 1), S.MASK_ID, device=DEV, dtype=torch.long)), dim=1)
        Lk = in_len + s + 1
        emb, _ = O.embeddings(pd, vf, vp, x, dv[3][:, :Lk], 100, position_ids=dv[4][:, :Lk])
        hs = O.encoder(pd, emb, O.extended_attention_mask(dv[5][:, :Lk, :Lk], torch.float32), 12)
        logp = torch.log_softmax(O.lm_head(pd, hs[-1][:, -1:, :])[:, 0], dim=-1)
        ref = torch.gather(logp, 1, ids[:, s:s + 1])[:, 0]
        assert float((ref - lps[:, s]).abs().max()) < 3e-2, (s, ref, lps[:, s])


# ------------------------------------------------------------------------------------------------
# beam search (modeling.py:1255-1494)
# ------------------------------------------------------------------------------------------------
from oracle.make_golden import BEAM_CASES                  # noqa: E402  (pure data)


@pytest.mark.parametrize("rows,V,Kb", [(5, 1000, 3), (64, 28996, 5), (2, 50, 50), (7, 28996, 8), (3, 100, 16), (2, 17, 16), (1, 300, 1)])
def test_logsoftmax_topk(rows, V, Kb):
    Vp = (V + 63) // 64 * 64
    g = torch.Generator().manual_seed(V + Kb)
    logits = (torch.randn(rows, Vp, generator=g) * 3).half().to(DEV)
    logits[:, V:] = 100.0
    forbid = (torch.rand(rows, V, generator=g) < 0.02).to(torch.uint8).to(DEV)
    eos = 7
    for fb, block in ((None, False), (forbid, True)):
        sc = torch.zeros(rows, Kb, dtype=torch.float32, device=DEV)
        ids = torch.zeros(rows, Kb, dtype=torch.long, device=DEV)
        K.logsoftmax_topk(logits, Vp, rows, V, Kb, sc, ids, forbid=fb, eos_id=eos, block_eos=block)
        ref = torch.log_softmax(logits[:, :V].float(), dim=-1)
        if fb is not None:
            ref = ref + fb.float() * -10000.0
        if block:
            ref[:, eos] = -10000.0
        es, ei = torch.topk(ref, Kb, dim=-1)
        assert float((sc - es).abs().max()) < 2e-4
        # ids may differ from torch.topk only inside exact ties: compare through the values they select
        assert float((torch.gather(ref, 1, ids) - es).abs().max()) < 2e-4
        assert all(len(set(r)) == Kb for r in ids.tolist())


def test_beam_select_and_kv_gather():
    B, Kb, eos = 3, 4, 9
    g = torch.Generator().manual_seed(5)
    kk_s = torch.randn(B * Kb, Kb, generator=g).to(DEV)
    kk_i = torch.randint(0, 30, (B * Kb, Kb), generator=g).to(DEV)
    last_tot = torch.randn(B, Kb, generator=g).to(DEV)
    last_eos = (torch.rand(B, Kb, generator=g) < 0.3).float().to(DEV)
    outs = [torch.zeros(B, Kb, device=DEV, dtype=dt) for dt in (torch.float32, torch.long, torch.long, torch.float32)]
    src = torch.zeros(B * Kb, dtype=torch.long, device=DEV)
    nxt = torch.zeros(B * Kb, 2, dtype=torch.long, device=DEV)
    K.beam_select(kk_s, kk_i, last_tot, last_eos, *outs, src, nxt[:, 0], B, Kb, False, eos)
    cand = (kk_s.view(B, Kb, Kb) + (last_eos * -10000.0 + last_tot).unsqueeze(-1)).reshape(B, Kb * Kb)
    es, sel = torch.topk(cand, Kb)
    assert torch.allclose(outs[0], es) and torch.equal(outs[2], sel // Kb)
    assert torch.equal(outs[1], torch.gather(kk_i.view(B, Kb * Kb), 1, sel))
    assert torch.equal(outs[3], (outs[1] == eos).float())
    assert torch.equal(src, (torch.arange(B, device=DEV).unsqueeze(1) * Kb + outs[2]).reshape(-1))
    assert torch.equal(nxt[:, 0], outs[1].reshape(-1)) and int(nxt[:, 1].abs().sum()) == 0
    # first frame: the K candidates of the single row, pointers 0, cache rows = sample index
    kk_s = torch.sort(kk_s, dim=1, descending=True)[0]          # what vlp_logsoftmax_topk hands over: best first
    K.beam_select(kk_s, kk_i, None, None, *outs, src, nxt[:, 0], B, Kb, True, eos)
    assert torch.equal(outs[0], kk_s[:B]) and torch.equal(outs[1], kk_i[:B]) and int(outs[2].abs().sum()) == 0
    assert torch.equal(src, torch.arange(B, device=DEV).repeat_interleave(Kb))
    # cache rows follow src
    R, Lcap, E = B * Kb, 12, 64
    a = torch.randn(R, Lcap, E, generator=g).half().to(DEV)
    b = torch.zeros_like(a)
    idx = torch.randint(0, R, (R,), generator=g).to(DEV)
    K.kv_gather(a, Lcap, b, Lcap, idx, R, 3, 9, E)
    exp = torch.zeros_like(a)
    exp[:, 3:9] = a[idx][:, 3:9]
    assert torch.equal(b, exp)


def compare_beams(tr, g, Kb, n_frames):
    """Frame by frame per sample: words / back pointers must equal the reference's as long as the reference's own margin between
    the K-th kept and the best rejected continuation exceeds the fp16 noise; scores within tolerance up to there."""
    full = 0
    B = g["wids"].shape[0]
    for b in range(B):
        ok_frames = 0
        for f in range(n_frames):
            rs, ms = g["scores"][b, f], tr["scores"][b, f].cpu().numpy()
            same = np.array_equal(g["wids"][b, f], tr["wids"][b, f].cpu().numpy()) and np.array_equal(g["ptrs"][b, f], tr["ptrs"][b, f].cpu().numpy())
            if not same:
                # a different ORDER or membership is only acceptable next to a near-tie: either the K-th/K+1-th margin, or two kept
                # hypotheses closer than the noise
                kept_gap = np.min(np.abs(np.diff(np.sort(rs)))) if Kb > 1 else np.inf
                assert min(g["margins"][b, f], kept_gap) < MARGIN_TOL, "sample %d frame %d: beams differ although margins are %g / %g" % (
                    b, f, g["margins"][b, f], kept_gap)
                break
            # cumulative log-probabilities of fp16 logits: every frame may add 2e-3 of the logit magnitude (north-star 1e-3 relative
            # per evaluation, two evaluations compared; one fp16 ulp of a logit near 16 is already 1.6e-2)
            tol = 2e-3 * float(g["logit_scale"]) * (f + 1) + 1e-3
            assert np.max(np.abs(rs - ms)) < tol, "sample %d frame %d scores %s vs %s (tol %g)" % (b, f, ms, rs, tol)
            ok_frames += 1
        if ok_frames == n_frames:
            full += 1
            L = g["pred_seq"].shape[1]
            assert np.array_equal(g["pred_seq"][b], tr["pred_seq"][b].cpu().numpy()[:L]), "sample %d: back-tracked sequence differs" % b
    return full


@pytest.mark.parametrize("name", list(BEAM_CASES.keys()))
def test_beam_search_vs_reference_fixture(name):
    mk, B, T, seed, dk = BEAM_CASES[name]
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    p = O.init_params(vocab_size=mk["vocab_size"], layers=mk["layers"], tasks=mk["tasks"], seed=mk["seed"], std=mk["std"])
    inp = decode_inputs(B, T, seed)
    if not np.allclose(decode_fingerprint(p, inp), g["fingerprint"], rtol=1e-9, atol=0):
        pytest.skip("RNG stream differs from the one that generated the fixture")
    cfg = BertConfig(mk["vocab_size"], num_hidden_layers=mk["layers"], type_vocab_size=6)
    m = BertForSeq2SeqDecoder(cfg, mask_word_id=S.MASK_ID, eos_id=S.SEP_ID, enable_butd=True, len_vis_input=100, **dk)
    sd = dict(p)
    sd["cls.predictions.decoder.weight"] = p["bert.embeddings.word_embeddings.weight"]
    m.load_state_dict(sd, strict=True)
    m = m.half().to(DEV).eval()
    img, vis_pe, input_ids, token_type, pos, am = [t.to(DEV) for t in inp]
    tr = m(img.half(), vis_pe.half(), input_ids, token_type, pos, am, task_idx=None)
    Kb, out_len = dk["search_beam_size"], token_type.shape[1]
    assert tr["pred_seq"].shape == (B, out_len) and tr["scores"].shape == (B, out_len, Kb) and tr["wids"].dtype == torch.long
    assert tr["scores"].is_cuda
    full = compare_beams(tr, g, Kb, T)
    if name == "beam_2l_K3":          # fixture chosen with every margin > 0.03: the whole search must be identical
        assert full == B
    for rep in range(3):                  # call 2 records the launch plans (when no n-gram blocking is active), later calls replay them
        tr2 = m(img.half().clone(), vis_pe.half().clone(), input_ids.clone(), token_type.clone(), pos.clone(), am.clone())
        assert all(torch.equal(tr[k], tr2[k]) for k in tr), rep
