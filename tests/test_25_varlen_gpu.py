"""GPU: the padding-free (packed) training step (Engine.varlen / VLP_VARLEN=1) against the dense step of the SAME engine and against the
oracle.  A sample's positions past its last token are inert in the reference's computation (attended by no query, read by no loss:
seq2seq_loader.py:295-304, modeling.py:289-298), so dropping them must leave losses, logits, every gradient and the updated parameters
unchanged: the forward bit for bit (row-wise kernels + per-sample attention see the same numbers, dropout hashes keep the logical
element), the parameter gradients up to the fp32 summation order of sums over rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from oracle import vlp_oracle as O                                      # noqa: E402 (checker)
from vlp_amd import _lib as K                                           # noqa: E402
from vlp_amd import synthetic as S                                      # noqa: E402
from vlp_amd.input_prep import MaskSpec                                 # noqa: E402
from vlp_amd.modeling import BertConfig, BertForPreTrainingLossMask     # noqa: E402
from vlp_amd.optimization_fp16 import FP16_Optimizer_State, FusedAdam   # noqa: E402

DEV = torch.device("cuda:0")
ND = ["bias", "LayerNorm.bias", "LayerNorm.weight"]


def packing(lens, L):
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    row_off = torch.from_numpy(off).to(DEV)
    row_map = torch.full((int(off[-1]),), -1, dtype=torch.int32, device=DEV)
    K.rowmap_build(row_off, len(lens), L, row_map)
    want = np.concatenate([b * L + np.arange(n) for b, n in enumerate(lens)]).astype(np.int32)
    assert np.array_equal(row_map.cpu().numpy(), want)
    return row_off, row_map, int(off[-1])


def test_rowmap_pack_unpack_gather_scatter_vqa():
    B, L, H, P, Nv = 5, 40, 64, 3, 10
    lens = [40, 13, 27, 12, 33]
    row_off, row_map, Mp = packing(lens, L)
    g = torch.Generator(device=DEV).manual_seed(0)
    dense = torch.randn(B * L, H, device=DEV, generator=g).half()
    packed = torch.empty(Mp, H, device=DEV, dtype=torch.float16)
    K.rows_pack(dense, row_map, Mp, packed, H)
    assert torch.equal(packed, dense[row_map.long()])
    back = torch.zeros_like(dense)
    K.rows_unpack(packed, row_map, Mp, back, H)
    keep = torch.zeros(B * L, dtype=torch.bool, device=DEV)
    keep[row_map.long()] = True
    assert torch.equal(back[keep], dense[keep]) and float(back[~keep].abs().max()) == 0.0
    pos = torch.stack([torch.randint(0, n, (P,)) for n in lens]).to(DEV)
    a, b = torch.empty(B * P, H, device=DEV, dtype=torch.float16), torch.empty(B * P, H, device=DEV, dtype=torch.float16)
    K.gather_rows(dense, H, pos, a, H, B, P, L, H)
    K.gather_rows(packed, H, pos, b, H, B, P, L, H, row_off=row_off)
    assert torch.equal(a, b)
    d0, d1 = torch.zeros_like(dense), torch.zeros_like(packed)
    K.scatter_add_rows(a, H, pos, d0, H, B, P, L, H)
    K.scatter_add_rows(a, H, pos, d1, H, B, P, L, H, row_off=row_off)
    assert torch.equal(d0[row_map.long()], d1)
    e0, e1 = torch.empty(B, H, device=DEV, dtype=torch.float16), torch.empty(B, H, device=DEV, dtype=torch.float16)
    K.vqa_mul_fwd(dense, e0, B, L, Nv, H)
    K.vqa_mul_fwd(packed, e1, B, L, Nv, H, row_off=row_off)
    assert torch.equal(e0, e1)
    K.vqa_mul_bwd(dense, e0, d0, B, L, Nv, H)
    K.vqa_mul_bwd(packed, e0, d1, B, L, Nv, H, row_off=row_off)
    assert torch.equal(d0[row_map.long()], d1)


@pytest.mark.parametrize("N,Kd,variant", [(768, 768, 77), (768, 3072, 77), (768, 768, 27), (768, 768, 10)])
def test_gemm_dropout_masks_follow_the_logical_row(N, Kd, variant):
    """vlp_gemm_nt(row_map): the packed run's output rows equal the dense run's rows bit for bit with dropout ON."""
    B, L = 6, 167
    lens = [167, 110, 139, 103, 150, 121]
    row_off, row_map, Mp = packing(lens, L)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = (torch.randn(B * L, Kd, device=DEV, generator=g) * 0.5).half()
    w = (torch.randn(N, Kd, device=DEV, generator=g) * 0.05).half()
    bias, res = torch.randn(N, device=DEV, generator=g).half(), torch.randn(B * L, N, device=DEV, generator=g).half()
    y0 = torch.empty(B * L, N, device=DEV, dtype=torch.float16)
    K.gemm_nt(x, w, y0, B * L, N, Kd, bias=bias, residual=res, dropout_p=0.1, seed=77, rng_stream=5, variant=variant)
    xp, rp = x[row_map.long()].contiguous(), res[row_map.long()].contiguous()
    y1 = torch.empty(Mp, N, device=DEV, dtype=torch.float16)
    K.gemm_nt(xp, w, y1, Mp, N, Kd, bias=bias, residual=rp, dropout_p=0.1, seed=77, rng_stream=5, variant=variant, row_map=row_map)
    assert torch.equal(y0[row_map.long()], y1)
    y2 = torch.empty_like(y1)
    K.gemm_nt(xp, w, y2, Mp, N, Kd, bias=bias, residual=rp, dropout_p=0.1, seed=77, rng_stream=5, variant=variant)      # without the map: other masks
    assert not torch.equal(y1, y2)


def test_layernorm_dropout_masks_follow_the_logical_row():
    B, L, H = 4, 123, 768
    lens = [123, 104, 110, 117]
    row_off, row_map, Mp = packing(lens, L)
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(B * L, H, device=DEV, generator=g).half()
    gam, bet = torch.randn(H, device=DEV, generator=g).half(), torch.randn(H, device=DEV, generator=g).half()
    y0, y1 = torch.empty_like(x), torch.empty(Mp, H, device=DEV, dtype=torch.float16)
    m0, r0 = torch.empty(B * L, device=DEV), torch.empty(B * L, device=DEV)
    m1, r1 = torch.empty(Mp, device=DEV), torch.empty(Mp, device=DEV)
    K.layernorm_fwd(x, gam, bet, y0, B * L, H, m0, r0, dropout_p=0.1, seed=9, rng_stream=1000)
    xp = x[row_map.long()].contiguous()
    K.layernorm_fwd(xp, gam, bet, y1, Mp, H, m1, r1, dropout_p=0.1, seed=9, rng_stream=1000, row_map=row_map)
    assert torch.equal(y0[row_map.long()], y1) and torch.equal(m0[row_map.long()], m1)
    dy = torch.randn(B * L, H, device=DEV, generator=g).half()
    ws = torch.empty(K.layernorm_bwd_workspace_bytes(H), device=DEV, dtype=torch.uint8)
    outs = []
    for (xx, dd, mm, rr, M, rmap) in ((x, dy, m0, r0, B * L, None), (xp, dy[row_map.long()].contiguous(), m1, r1, Mp, row_map)):
        dx, dxd = torch.empty(M, H, device=DEV, dtype=torch.float16), torch.empty(M, H, device=DEV, dtype=torch.float16)
        dg, db = torch.empty(H, device=DEV, dtype=torch.float16), torch.empty(H, device=DEV, dtype=torch.float16)
        K.layernorm_bwd(dd, xx, gam, mm, rr, dx, dg, db, M, H, ws, dx_drop=dxd, dy_drop=(0.1, 9, 1000), out_drop=(0.1, 9, 35), row_map=rmap)
        outs.append((dx, dxd))
    assert torch.equal(outs[0][0][row_map.long()], outs[1][0]) and torch.equal(outs[0][1][row_map.long()], outs[1][1])


@pytest.mark.parametrize("L,drop", [(167, 0.1), (123, 0.0), (60, 0.1)])
def test_attention_packed_rows_equal_dense(L, drop):
    """vlp_attn_fwd / vlp_attn_bwd with row_off: context rows, lse and dQ / dK / dV of the kept positions equal the dense launch bit for
    bit (the dropped positions carry dO = 0 in the dense run, as they do in a training step)."""
    B, A, H, Nv = 5, 12, 768, 30
    rng = np.random.RandomState(3)
    nb = rng.randint(1, L - Nv - 3, size=B)
    nb[0] = L - Nv - 3
    modes = [bool(rng.rand() < 0.6) for _ in range(B)]
    spec = MaskSpec.from_lengths(Nv, nb.tolist(), modes, device=DEV)
    lens = spec.lens_host
    row_off, row_map, Mp = packing(lens, L)
    Lp = (L + 31) // 32 * 32
    maskb, maskt = torch.empty(B, L, Lp, dtype=torch.uint8, device=DEV), torch.empty(B, Lp, Lp, dtype=torch.uint8, device=DEV)
    K.mask_build(spec.second_st, spec.second_end, spec.is_s2s, maskb, B, L, Lp, out_t=maskt)
    g = torch.Generator(device=DEV).manual_seed(4)
    qkv = (torch.randn(B * L, 3 * H, device=DEV, generator=g) * 0.7).half()
    dctx = (torch.randn(B * L, H, device=DEV, generator=g) * 0.3).half()
    keep = torch.zeros(B * L, dtype=torch.bool, device=DEV)
    keep[row_map.long()] = True
    dctx[~keep] = 0                                           # padding queries receive no gradient in a training step
    scale = 0.125
    ctx0, lse0 = torch.zeros(B * L, H, device=DEV, dtype=torch.float16), torch.zeros(B, A, L, device=DEV)
    K.attn_fwd(qkv, maskb, ctx0, lse0, B, L, A, scale, dropout_p=drop, seed=11, rng_stream=17)
    qkv_p, dctx_p = qkv[row_map.long()].contiguous(), dctx[row_map.long()].contiguous()
    ctx1, lse1 = torch.zeros(Mp, H, device=DEV, dtype=torch.float16), torch.zeros(B, A, L, device=DEV)
    K.attn_fwd(qkv_p, maskb, ctx1, lse1, B, L, A, scale, dropout_p=drop, seed=11, rng_stream=17, row_off=row_off)
    assert torch.equal(ctx0[row_map.long()], ctx1)
    kept_q = keep.view(B, 1, L).expand(B, A, L)
    assert torch.equal(lse0[kept_q], lse1[kept_q])
    dq0, dl0 = torch.zeros(B * L, 3 * H, device=DEV, dtype=torch.float16), torch.zeros(B, A, L, device=DEV)
    K.attn_bwd(qkv, maskb, maskt, ctx0, dctx, lse0, dq0, dl0, B, L, A, scale, dropout_p=drop, seed=11, rng_stream=17)
    dq1, dl1 = torch.zeros(Mp, 3 * H, device=DEV, dtype=torch.float16), torch.zeros(B, A, L, device=DEV)
    K.attn_bwd(qkv_p, maskb, maskt, ctx1, dctx_p, lse1, dq1, dl1, B, L, A, scale, dropout_p=drop, seed=11, rng_stream=17, row_off=row_off)
    assert torch.equal(dq0[row_map.long()], dq1)
    assert float(dq0[~keep].abs().max()) == 0.0              # the dense run's gradient on the dropped positions is exactly zero


def _model(tasks, drop, layers=2, seed=3, vocab=1024):
    p = O.init_params(vocab_size=vocab, layers=layers, tasks=tasks, seed=seed)
    cfg = BertConfig(vocab, num_hidden_layers=layers, type_vocab_size=6, hidden_dropout_prob=drop, attention_probs_dropout_prob=drop)
    m = BertForPreTrainingLossMask(cfg, enable_butd=True, len_vis_input=100, tasks=tasks, allow_random_fc7=True)
    sd = dict(p)
    sd["cls.predictions.decoder.weight"] = p["bert.embeddings.word_embeddings.weight"]
    m.load_state_dict(sd, strict=True)
    return m.half().to(DEV).train(), p


def _groups(model):
    named = list(model.named_parameters())
    return [{"params": [q for n, q in named if not any(x in n for x in ND)], "weight_decay": 0.01},
            {"params": [q for n, q in named if any(x in n for x in ND)], "weight_decay": 0.0}]


def _step(model, opt, b, mir=False):
    lt = model(b.img, b.vis_pe, b.input_ids, b.segment_ids, b.input_mask, b.lm_label_ids, b.ans_labels, b.is_next, masked_pos=b.masked_pos,
               masked_weights=b.masked_weights, task_idx=b.task_idx, vis_masked_pos=b.vis_masked_pos, mask_image_regions=mir, drop_worst_ratio=0)
    opt.backward(lt[0] + lt[1] + lt[2])
    return lt


@pytest.mark.parametrize("tasks,max_len_b,s2s_prob,vis_mask_prob,spec", [("img2txt", 64, 1.0, 0.0, False), ("img2txt", 20, 0.5, 0.0, True),
                                                                         ("vqa2", 64, 0.0, 0.0, False), ("img2txt", 20, 0.75, 0.25, False)])
def test_packed_step_equals_dense_step(tasks, max_len_b, s2s_prob, vis_mask_prob, spec):
    """Same model, same batch, dropout 0.1, dense vs packed: loss and logits BIT-equal (so the dropout masks are), every gradient tensor
    within fp32-summation noise (<= 2e-4 rel-L2; the position / type / word tables and everything upstream of the first row-sum
    bit-equal), and the parameters after one FusedAdam step likewise."""
    B = 8
    raw = S.make_batch(B, max_len_b=max_len_b, vocab_size=1024, max_pred=3 if tasks != "vqa2" else 1, s2s_prob=s2s_prob, tasks=tasks, seed=21,
                       vis_mask_prob=vis_mask_prob)
    batch = S.batch_to(raw, DEV, half=True)
    if spec:      # the loader's form: lengths instead of the dense mask (host lengths ride along: no read-back)
        nb = [int(raw.input_mask[i].any(dim=0).sum()) - 103 for i in range(B)]
        ms = MaskSpec.from_lengths(100, nb, [int(t) == 3 for t in raw.task_idx], device=DEV)
        assert torch.equal(ms.dense(raw.input_mask.shape[1]).cpu(), raw.input_mask)
        batch = batch._replace(input_mask=ms)
    res = {}
    for mode in ("dense", "packed"):
        model, _ = _model(tasks, 0.1)
        eng = model.engine
        eng.varlen = mode == "packed"
        opt = FP16_Optimizer_State(FusedAdam(_groups(model), lr=1e-3, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True,
                                   dynamic_loss_args={"init_scale": 1.0 if tasks == "vqa2" else 2.0 ** 10})      # (the VQA loss is ~2 000: BCE x 3 129)
        lt = _step(model, opt, batch, mir=vis_mask_prob > 0)
        torch.cuda.synchronize()
        L = raw.input_ids.shape[1]
        if mode == "packed":
            assert eng.last_packed_rows is not None and eng.last_packed_rows < B * L
            lens = [int(raw.input_mask[i].any(dim=0).nonzero().max()) + 1 for i in range(B)]
            assert eng.last_packed_rows == sum(lens)
        else:
            assert eng.last_packed_rows is None
        logits = (model.last_vqa_logits if tasks == "vqa2" else model.last_mlm_logits).clone()
        grads = {n: q.grad.detach().float().clone() for n, q in model.named_parameters()}
        for g_ in opt.param_groups:
            g_["lr"] = 1e-3
        opt.step()
        torch.cuda.synchronize()
        assert not opt.overflow
        params = {n: q.detach().float().clone() for n, q in model.named_parameters()}
        masters = [t.clone() for t in opt.fp32_groups_flat]
        res[mode] = (torch.stack([x.detach().float().reshape(()) for x in lt]).clone(), logits, grads, params, masters)
    l0, g0, gr0, p0, m0 = res["dense"]
    l1, g1, gr1, p1, m1 = res["packed"]
    assert torch.equal(l0, l1), (l0, l1)
    assert torch.equal(g0, g1)
    worst = ("", 0.0)
    for n in gr0:
        d, ref = float((gr0[n] - gr1[n]).norm()), float(gr0[n].norm())
        rel = d / ref if ref > 0 else d
        if rel > worst[1]:
            worst = (n, rel)
        assert rel <= 2e-4, (n, rel)
    for n in ("bert.embeddings.position_embeddings.weight", "bert.embeddings.token_type_embeddings.weight"):
        assert torch.equal(gr0[n], gr1[n]), n              # summed in the dense geometry from bit-equal row gradients
    for n in p0:
        # one Adam step of lr 1e-3 moves a weight by ~3e-3 (no bias correction: m / sqrt(v) = 0.1 / sqrt(0.001)); the two runs' updates agree to
        # a fraction of a percent of that even where a gradient element sits at the fp16 noise level (measured 7.6e-6)
        # the fp16 model copy: the masters' distance (below) + one rounding
        assert bool(((p0[n] - p1[n]).abs() <= 3e-5 + 2.0 ** -10 * torch.maximum(p0[n].abs(), p1[n].abs())).all()), n
    for a, b in zip(m0, m1):
        assert float((a - b).abs().max()) <= 3e-5, float((a - b).abs().max())
    print("packed vs dense: worst gradient tensor %s rel-L2 %.2e" % worst)


def test_packed_forward_vs_oracle():
    """Dropout 0: the packed run against the oracle's fp32 evaluation (smoke()'s criteria); its hidden states are the dense run's on the
    kept positions, bit for bit."""
    B = 4
    raw = S.make_batch(B, max_len_b=20, vocab_size=1024, max_pred=3, seed=5)
    batch = S.batch_to(raw, DEV, half=True)
    out = {}
    for mode in ("dense", "packed"):
        model, p = _model("img2txt", 0.0)
        model.engine.varlen = mode == "packed"
        lt = model(batch.img, batch.vis_pe, batch.input_ids, batch.segment_ids, batch.input_mask, batch.lm_label_ids, batch.ans_labels,
                   batch.is_next, masked_pos=batch.masked_pos, masked_weights=batch.masked_weights, task_idx=batch.task_idx, drop_worst_ratio=0)
        eng = model.engine
        ws = eng._ws[next(iter(eng._ws))]
        hs = ws["layers"][1]["x2"]
        out[mode] = (float(lt[0].detach()), model.last_mlm_logits.float().clone(), hs.clone(), eng.last_packed_rows)
    want = O.forward_pretraining_loss_mask({k: v.to(DEV) for k, v in p.items()}, S.batch_to(raw, DEV), tasks="img2txt")
    for mode in out:
        got, logits = out[mode][0], out[mode][1]
        assert abs(got - float(want["mlm_loss"])) < 3e-3 * abs(float(want["mlm_loss"]))
        assert float((logits - want["mlm_logits"]).abs().max() / want["mlm_logits"].abs().max()) < 4e-3
    assert out["dense"][0] == out["packed"][0] and torch.equal(out["dense"][1], out["packed"][1])
    Mp = out["packed"][3]
    L = raw.input_ids.shape[1]
    lens = [int(raw.input_mask[i].any(dim=0).nonzero().max()) + 1 for i in range(B)]
    rows = torch.cat([b * L + torch.arange(n) for b, n in enumerate(lens)]).to(DEV)
    assert Mp == rows.numel() and torch.equal(out["dense"][2][rows], out["packed"][2][:Mp])


def test_eval_and_no_grad_forwards_stay_dense():
    model, _ = _model("vqa2", 0.0)
    model.engine.varlen = True
    raw = S.make_batch(4, max_len_b=20, vocab_size=1024, max_pred=1, tasks="vqa2", seed=6)
    b = S.batch_to(raw, DEV, half=True)
    model.eval()
    with torch.no_grad():
        model(b.img, b.vis_pe, b.input_ids, b.segment_ids, b.input_mask, vqa_inference=True)
    assert model.engine.last_packed_rows is None


@pytest.mark.skipif(not K.lab_build(), reason="investigation kernel: needs a -DVLP_LAB_BUILD library (python -m vlp_amd.build --lab, VLP_HIP_LIB=...)")
@pytest.mark.parametrize("B,L,packed,drop", [(64, 167, False, 0.1), (64, 167, True, 0.1), (130, 150, False, 0.0), (3, 192, False, 0.1), (30, 129, True, 0.0)])
def test_attention_forward_streaming_kernel_equals_the_per_item_kernel(B, L, packed, drop, monkeypatch):
    """The persistent streaming forward (one 12-wave workgroup per CU, K / V of the next item prefetched through three LDS buffers, tiles
    handed out by an LDS counter; 129 <= L <= 192) runs the same tile code as the one-workgroup-per-(batch, head) kernel: context rows and
    lse must be BIT-equal -- with several items per workgroup (B x heads > CUs: buffers reused), a grid smaller than the CU count, packed
    rows, dropout on and off."""
    A, H, Nv = 12, 768, 100
    rng = np.random.RandomState(B + L)
    nb = rng.randint(1, L - Nv - 2, size=B)
    nb[0] = L - Nv - 3
    modes = [bool(rng.rand() < 0.7) for _ in range(B)]
    spec = MaskSpec.from_lengths(Nv, nb.tolist(), modes, device=DEV)
    Lp = (L + 31) // 32 * 32
    maskb = torch.empty(B, L, Lp, dtype=torch.uint8, device=DEV)
    K.mask_build(spec.second_st, spec.second_end, spec.is_s2s, maskb, B, L, Lp)
    g = torch.Generator(device=DEV).manual_seed(5)
    if packed:
        row_off, row_map, M = packing(spec.lens_host, L)
    else:
        row_off, M = None, B * L
    qkv = (torch.randn(M, 3 * H, device=DEV, generator=g) * 0.7).half()
    outs = []
    for stream in ("0", "1"):
        monkeypatch.setenv("VLP_ATTN_FWD_STREAM", stream)
        ctx, lse = torch.zeros(M, H, device=DEV, dtype=torch.float16), torch.zeros(B, A, L, device=DEV)
        for _ in range(2):          # twice: no state is left behind
            K.attn_fwd(qkv, maskb, ctx, lse, B, L, A, 0.125, dropout_p=drop, seed=3, rng_stream=9, row_off=row_off)
        outs.append((ctx, lse))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[1][0].float().abs().max()) > 0
