"""BASELINE.json's FULL size (BERT-base 12 layers, vocab 28 996, 100 regions, seq 64 -> L = 167, batch 64):
(1) size-independent properties -- bitwise determinism, exact equivariance under a permutation of the samples, the expected loss
    of an untrained model (ln V), linearity of the backward pass in the loss scale;
(2) parity with the oracle run in fp32 ON THE DEVICE (second half of this file): logits, loss and every gradient tensor
    element-wise, for the COCO (seq2seq), Conceptual-Captions (mixed masks) and VQA shapes, and the logits under every GEMM variant."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from vlp_amd import synthetic as S                                # noqa: E402
from vlp_amd.modeling import BertConfig, BertForPreTrainingLossMask   # noqa: E402

DEV = torch.device("cuda:0")
V, B = 28996, 64


@pytest.fixture(scope="module")
def model():
    torch.manual_seed(0)
    cfg = BertConfig(V, num_hidden_layers=12, type_vocab_size=6, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    return BertForPreTrainingLossMask(cfg, enable_butd=True, len_vis_input=100, tasks="img2txt", allow_random_fc7=True).half().to(DEV).eval()


def run(m, b, scale=1.0, backward=False):
    losses = m(b.img, b.vis_pe, b.input_ids, b.segment_ids, b.input_mask, b.lm_label_ids, b.ans_labels, b.is_next, masked_pos=b.masked_pos,
               masked_weights=b.masked_weights, task_idx=b.task_idx, vis_masked_pos=b.vis_masked_pos, mask_image_regions=False, drop_worst_ratio=0.0)
    if backward:
        m.engine.zero_grad()
        (losses[0] * scale).sum().backward()
    torch.cuda.synchronize()
    return losses[0].detach().clone(), m.last_mlm_logits.detach().clone()


def test_full_size_padding_free_step_equals_dense(model):
    """BASELINE size (B = 64, L = 167, 12 layers), the SAME model object run dense and packed (Engine.varlen): with dropout 0 and with
    dropout 0.1 the logits and the loss are BIT-equal (so every parity statement about the dense logits holds for the packed step, and
    the dropout masks are the dense run's), every gradient tensor agrees to fp32-summation noise (<= 2e-4 rel-L2)."""
    eng = model.engine
    batch = S.batch_to(S.make_batch(B, max_len_b=64, vocab_size=V, max_pred=3, s2s_prob=0.75, seed=7), DEV, half=True)
    cfg = model.config
    try:
        for pdrop in (0.0, 0.1):
            cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = pdrop
            model.train(pdrop > 0)
            res = []
            for packed in (False, True):
                eng.varlen = packed
                eng.step_seed = 41                       # same dropout stream position for both runs
                loss, logits = run(model, batch, scale=1024.0, backward=True)
                assert (eng.last_packed_rows is not None) == packed
                if packed:
                    L = batch.input_ids.shape[1]
                    assert eng.last_packed_rows == int(sum(int(batch.input_mask[i].any(dim=0).nonzero().max()) + 1 for i in range(B))) < B * L
                res.append((loss, logits, {n: q.grad.detach().float().clone() for n, q in model.named_parameters()}))
            (l0, g0, gr0), (l1, g1, gr1) = res
            assert torch.equal(l0, l1) and torch.equal(g0, g1), pdrop
            worst = max((float((gr0[n] - gr1[n]).norm()) / max(float(gr0[n].norm()), 1e-30), n) for n in gr0 if float(gr0[n].norm()) > 0)
            print("full size, dropout %.1f: packed rows %d of %d, worst gradient tensor rel-L2 %.2e (%s)" % (pdrop, eng.last_packed_rows, B * 167, worst[0], worst[1]))
            assert worst[0] <= 2e-4, worst
    finally:
        eng.varlen = False
        cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = 0.0
        model.eval()


def permuted(b, perm):
    return type(b)(*[t[perm] if torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == B else t for t in b])


def test_full_size_properties(model):
    batch = S.batch_to(S.make_batch(B, max_len_b=64, vocab_size=V, max_pred=3, s2s_prob=0.75, seed=7), DEV, half=True)
    loss1, logits1 = run(model, batch)
    loss2, logits2 = run(model, batch)
    assert logits1.shape == (B, 3, V)
    # (1) bitwise reproducible
    assert torch.equal(logits1, logits2) and torch.equal(loss1, loss2)
    # (2) an untrained model predicts ~uniformly: loss ~ ln V (the reference's own sanity value, SURVEY.md 8c: 10.59 with its init)
    assert abs(float(loss1) - math.log(V)) < 0.06 * math.log(V), float(loss1)
    # (3) samples are independent: permuting the batch permutes the logits EXACTLY (every row of every GEMM accumulates in the same order
    #     wherever it sits in the tile grid; attention works per (sample, head))
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(DEV)
    _, logits_p = run(model, permuted(batch, perm))
    assert torch.equal(logits_p, logits1[perm])
    # (4) the backward pass is linear in the upstream scale: doubling the loss scale doubles every gradient (fp16 grads: compare in
    #     relative L2, gradient entries near the fp16 subnormal range round differently)
    run(model, batch, scale=2048.0, backward=True)
    g1 = {k: v.float().clone() for k, v in model.engine.gflat.items()}
    run(model, batch, scale=4096.0, backward=True)
    for k, v in model.engine.gflat.items():
        rel = float((v.float() - 2.0 * g1[k]).norm() / (v.float().norm() + 1e-30))
        assert rel < 2e-3, (k, rel)
    # (5) and itself reproducible bit for bit
    g2 = {k: v.clone() for k, v in model.engine.gflat.items()}
    run(model, batch, scale=4096.0, backward=True)
    assert all(torch.equal(g2[k], model.engine.gflat[k]) for k in g2)


# =====================================================================================================================
# Full-size parity against the oracle ON THE DEVICE: B = 64, L = 167, V = 28 996, 12 layers.
# The oracle (oracle/vlp_oracle.py, pinned to the unmodified reference on CPU) runs in fp32 on the MI355X through torch --
# seconds, not hours -- so the ragged paths that only exist at full size (M = 10 688 = 83.5 x 128 rows, the 29 056-pitch LM head
# at R = 192, split-M wgrads over 10 688 rows) are compared with it element by element.
#   criterion (ii), asserted:  err(hip vs fp32 truth) <= err(reference-fp16 vs fp32 truth) + 1e-3        (max-rel on logits)
#   criterion (i), reported and bounded: hip vs reference-fp16 directly (two independent fp16 evaluations of a 12-layer net)
# Gradients: relative L2 error of EVERY parameter tensor against the fp32 truth, element-wise (not norms), with the
# reference-fp16 gradient's own error as the yardstick.  Both fp16 runs back-propagate a x4096 scaled loss (fp16 gradients of an
# unscaled loss underflow; the train loop scales by 65 536).
# =====================================================================================================================
import json      # noqa: E402
import os        # noqa: E402

from oracle import vlp_oracle as O      # noqa: E402  (checker only)

FULL_REPORT = {}


def _relmax(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def _relL2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _build(p, tasks, layers=12):
    cfg = BertConfig(V, num_hidden_layers=layers, type_vocab_size=6, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = BertForPreTrainingLossMask(cfg, enable_butd=True, len_vis_input=100, tasks=tasks, allow_random_fc7=True)
    sd = dict(p)
    sd["cls.predictions.decoder.weight"] = p["bert.embeddings.word_embeddings.weight"]
    m.load_state_dict(sd, strict=True)
    return m.half().to(DEV).eval()


def _hip(m, batch, scale=None):
    b = S.batch_to(batch, DEV, half=True)
    losses = m(b.img, b.vis_pe, b.input_ids, b.segment_ids, b.input_mask, b.lm_label_ids, b.ans_labels, b.is_next, masked_pos=b.masked_pos,
               masked_weights=b.masked_weights, task_idx=b.task_idx, vis_masked_pos=b.vis_masked_pos,
               mask_image_regions=b.vis_masked_pos.numel() > 0, drop_worst_ratio=0.0)
    if scale is not None:
        m.engine.zero_grad()
        ((losses[0] + losses[1] + losses[2]).sum() * scale).backward()
    torch.cuda.synchronize()
    return losses


def _oracle(p, batch, tasks, dtype, scale=None):
    pd = {k: v.to(DEV).to(dtype).clone().requires_grad_(scale is not None) for k, v in p.items()}
    b = S.batch_to(batch, DEV)
    mir = batch.vis_masked_pos.numel() > 0
    if scale is None:
        with torch.no_grad():
            return O.forward_pretraining_loss_mask(pd, b, tasks=tasks, mask_image_regions=mir), None
    out = O.forward_pretraining_loss_mask(pd, b, tasks=tasks, mask_image_regions=mir)
    (out["loss"].sum().float() * scale).backward()
    grads = {k: (None if t.grad is None else t.grad.float() / scale) for k, t in pd.items()}
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}, grads


FULL_CASES = {
    # BASELINE.json configs[1]/[2]: COCO captions fine-tune, all seq2seq masks
    "coco_s2s": dict(tasks="img2txt", s2s_prob=1.0, max_pred=3, seed=101, gscale=4096.0),
    # configs[3]: Conceptual Captions pre-training shape, per-sample Bernoulli(0.75) seq2seq / bidirectional masks
    "cc_mixed": dict(tasks="img2txt", s2s_prob=0.75, max_pred=3, seed=102, gscale=4096.0),
    # configs[4]: VQA 2.0 fine-tune, bidirectional, P = 1, answer-classifier head + BCE
    # (the BCE x 3129 loss of random labels is ~2 200: a x4 scale keeps loss x scale inside fp16 for the fp16 oracle)
    "vqa2": dict(tasks="vqa2", s2s_prob=0.0, max_pred=1, seed=103, gscale=4.0),
    # the pre-training shape with --vis_mask_prob 0.25 (mask_image_regions): 25 masked region rows per sample enter as zeros, their
    # mask columns are blocked, and the vis_pretext loss over the pooled output joins the sum (modeling.py:1049-1056, 1113-1131)
    "cc_vismask": dict(tasks="img2txt", s2s_prob=0.75, max_pred=3, seed=104, gscale=4096.0, vis_mask_prob=0.25),
}


@pytest.mark.parametrize("case", list(FULL_CASES))
def test_full_size_parity_vs_device_oracle(case):
    c = FULL_CASES[case]
    tasks, GSCALE = c["tasks"], c["gscale"]
    p = O.init_params(vocab_size=V, layers=12, tasks=tasks, seed=c["seed"])
    batch = S.make_batch(B, max_len_b=64, vocab_size=V, max_pred=c["max_pred"], s2s_prob=c["s2s_prob"], tasks=tasks, seed=c["seed"] + 7,
                         vis_mask_prob=c.get("vis_mask_prob", 0.0))
    assert batch.input_ids.shape == (B, 167)
    m = _build(p, tasks)
    losses = _hip(m, batch, scale=GSCALE)
    truth, gt = _oracle(p, batch, tasks, torch.float32, scale=1.0)
    ref16, g16 = _oracle(p, batch, tasks, torch.float16, scale=GSCALE)
    with O.rounding("sum_order"):            # the reference's arithmetic once more, every contraction summed in another order
        ref16b, _ = _oracle(p, batch, tasks, torch.float16)
    key = "vqa_logits" if tasks == "vqa2" else "mlm_logits"
    t = truth[key].float()
    ours = (m.last_vqa_logits if tasks == "vqa2" else m.last_mlm_logits).float().reshape(t.shape)
    r16 = ref16[key].float().reshape(t.shape)
    rep = {"logits_hip_vs_fp32": _relmax(ours, t), "logits_ref16_vs_fp32": _relmax(r16, t), "logits_hip_vs_ref16": _relmax(ours, r16),
           "logits_ref16_vs_ref16_other_summation_order": _relmax(ref16b[key].float().reshape(t.shape), r16),
           "logits_relL2_hip_vs_fp32": _relL2(ours, t), "logits_relL2_ref16_vs_fp32": _relL2(r16, t)}
    lt = float(truth["loss"].sum())
    lh = float((losses[0] + losses[1] + losses[2]).sum())
    rep.update(loss_fp32=lt, loss_hip=lh, loss_ref16=float(ref16["loss"].sum()))
    if c.get("vis_mask_prob"):
        rep.update(pretext_loss_fp32=float(truth["vis_pretext_loss"]), pretext_loss_hip=float(losses[1]), pretext_loss_ref16=float(ref16["vis_pretext_loss"]),
                   pooled_hip_vs_fp32=_relmax(m.last_pooled_output.float(), truth["pooled_output"].float()),
                   pooled_ref16_vs_fp32=_relmax(ref16["pooled_output"].float(), truth["pooled_output"].float()))
        assert rep["pooled_hip_vs_fp32"] <= rep["pooled_ref16_vs_fp32"] + 1e-3, rep
        assert abs(rep["pretext_loss_hip"] - rep["pretext_loss_fp32"]) <= abs(rep["pretext_loss_ref16"] - rep["pretext_loss_fp32"]) + 2e-3 * rep["pretext_loss_fp32"], rep
    # ---- gradients, element-wise, every tensor --------------------------------------------------------------------
    params = dict(m.named_parameters())
    unused = m.engine.unused_parameter_names()
    worst, worst_name, worst_excess, n_checked = 0.0, "", -1.0, 0
    per_tensor = {}
    for n, v in gt.items():
        if n == "cls.predictions.decoder.weight":
            continue
        if v is None:
            assert n in unused, n
            assert float(params[n].grad.float().abs().max()) == 0.0, n
            continue
        assert n not in unused, n
        mine = params[n].grad.float() / GSCALE
        e_h = float((mine.double() - v.double()).norm())
        e_r = float((g16[n].double() - v.double()).norm())
        nv = float(v.double().norm())
        if n.endswith("attention.self.key.bias"):
            # softmax is invariant to a shift of every key's score, so this gradient is EXACTLY zero in exact arithmetic: the fp32
            # "truth" is round-off.  Scale = the query-bias gradient of the same layer (the same column sum, over dQ instead of dK).
            nv = float(gt[n.replace("key.bias", "query.bias")].double().norm())
        rel_h, rel_r = e_h / (nv + 1e-30), e_r / (nv + 1e-30)
        per_tensor[n] = (rel_h, rel_r)
        n_checked += 1
        if rel_h > worst:
            worst, worst_name = rel_h, n
        worst_excess = max(worst_excess, rel_h - rel_r)
    rep.update(grad_tensors_checked=n_checked, grad_worst_relL2_hip=worst, grad_worst_tensor=worst_name,
               grad_worst_excess_over_ref16=worst_excess,
               grad_median_relL2_hip=sorted(x[0] for x in per_tensor.values())[len(per_tensor) // 2],
               grad_median_relL2_ref16=sorted(x[1] for x in per_tensor.values())[len(per_tensor) // 2])
    FULL_REPORT[case] = rep
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_fullsize.json", "w") as f:
        json.dump(FULL_REPORT, f, indent=1)
    with open("gpurun_out/parity_fullsize_grads_%s.json" % case, "w") as f:
        json.dump(per_tensor, f, indent=0)
    # ---- assertions ------------------------------------------------------------------------------------------------
    assert rep["logits_hip_vs_fp32"] <= rep["logits_ref16_vs_fp32"] + 1e-3, rep          # criterion (ii)
    # criterion (i): no farther from the reference-fp16 logits than the reference's own arithmetic is from itself under another
    # summation order, + the north-star 1e-3 (measured 2.75e-3 vs 2.20e-3 COCO, 3.49e-3 vs 2.86e-3 VQA; the decomposition test
    # below attributes the difference to rounding-point placement)
    assert rep["logits_hip_vs_ref16"] <= rep["logits_ref16_vs_ref16_other_summation_order"] + 1e-3, rep
    assert abs(lh - lt) <= 2e-3 * abs(lt), rep
    assert n_checked >= (208 if tasks == "img2txt" else 206) + (2 if c.get("vis_mask_prob") else 0), n_checked       # + the pooler
    for n, (rel_h, rel_r) in per_tensor.items():
        if n.endswith("attention.self.key.bias"):          # true value 0: bounded against the sibling query-bias gradient
            assert rel_h <= 5e-3, (n, rel_h, rel_r)         # measured <= 2.3e-3 of the query-bias gradient norm
            continue
        # measured (profiles/r02_parity_report.json): worst tensor 1.45e-2 / 1.70e-2 (vis_pe_embed.0.weight, img2txt / vqa2; the
        # reference's own fp16 arithmetic: 1.45e-2 / 1.69e-2), median 1.7e-3 / 3.5e-3, and never more than 4.5e-4 (img2txt) /
        # 1.9e-3 (vqa2, back-propagated with a x4 scale only) above the reference-fp16 error of the same tensor
        assert rel_h <= max(1.25 * rel_r, rel_r + (1e-3 if tasks == "img2txt" else 2.5e-3)), (n, rel_h, rel_r)
        assert rel_h <= 2e-2, (n, rel_h, rel_r)


@pytest.mark.parametrize("case", ["coco_s2s", "vqa2"])
def test_full_size_rounding_point_decomposition(case):
    """Where does the distance between the HIP path and the reference's fp16 arithmetic come from?  The oracle is
    evaluated in fp16 with the reference's op-by-op rounding, with ONE group of rounding points at a time moved to where the HIP
    kernels round (oracle.rounding flags), and with all of them moved.  The yardstick is the NOISE FLOOR of
    fp16 arithmetic on this network: the same evaluation with every Linear's contraction summed in another order (what a different
    GEMM tiling does; no rounding point moves) lands 1.7e-3 - 2.9e-3 max-rel (1.5e-3 - 1.9e-3 rel-L2) away from itself at 12
    layers -- no fp16 evaluation can be closer to another than that, in any norm.  Asserted: against the evaluation that rounds
    exactly where the HIP path rounds, the HIP logits sit AT that floor (measured 1.65e-3 vs 1.65e-3, 1.97e-3 vs 2.22e-3): nothing
    is left to explain beyond summation order; the whole extra distance to the reference-rounding logits (2.75e-3 / 3.49e-3) is
    rounding-point placement; and the placement costs no accuracy against the fp32 truth."""
    c = FULL_CASES[case]
    tasks = c["tasks"]
    p = O.init_params(vocab_size=V, layers=12, tasks=tasks, seed=c["seed"])
    batch = S.make_batch(B, max_len_b=64, vocab_size=V, max_pred=c["max_pred"], s2s_prob=c["s2s_prob"], tasks=tasks, seed=c["seed"] + 7)
    key = "vqa_logits" if tasks == "vqa2" else "mlm_logits"
    m = _build(p, tasks)
    with torch.no_grad():
        _hip(m, batch)
    truth, _ = _oracle(p, batch, tasks, torch.float32)
    t = truth[key].float()
    ours = (m.last_vqa_logits if tasks == "vqa2" else m.last_mlm_logits).float().reshape(t.shape)
    allk = "all flags = HIP rounding points"
    evals = [("reference rounding", ())] + [(f, (f,)) for f in O.ROUNDING_FLAGS] + [(allk, O.ROUNDING_FLAGS),
             ("reference rounding, other summation order", ("sum_order",)), (allk + ", other summation order", O.ROUNDING_FLAGS + ("sum_order",))]
    table, logits = {}, {}
    for name, flags in evals:
        with O.rounding(*flags):
            out, _ = _oracle(p, batch, tasks, torch.float16)
        logits[name] = out[key].float().reshape(t.shape)
    ref, hipr = logits["reference rounding"], logits[allk]
    for name, l in logits.items():
        table[name] = {"vs_fp32_truth": _relmax(l, t), "vs_hip": _relmax(l, ours), "vs_reference_rounding": _relmax(l, ref),
                       "relL2_vs_hip": _relL2(l, ours), "relL2_vs_fp32_truth": _relL2(l, t)}
    table["hip"] = {"vs_fp32_truth": _relmax(ours, t), "vs_reference_rounding": _relmax(ours, ref), "relL2_vs_fp32_truth": _relL2(ours, t)}
    # noise floor: the same rounding points, another fp32 summation order
    table["noise_floor_reference_rounding"] = {"maxrel": _relmax(logits["reference rounding, other summation order"], ref),
                                               "relL2": _relL2(logits["reference rounding, other summation order"], ref)}
    table["noise_floor_hip_rounding"] = {"maxrel": _relmax(logits[allk + ", other summation order"], hipr),
                                         "relL2": _relL2(logits[allk + ", other summation order"], hipr)}
    os.makedirs("gpurun_out", exist_ok=True)
    path = "gpurun_out/parity_rounding_decomposition.json"
    rep = json.load(open(path)) if os.path.exists(path) else {}
    rep[case] = table
    with open(path, "w") as f:
        json.dump(rep, f, indent=1)
    floor = table["noise_floor_hip_rounding"]
    # (a) same rounding points: the HIP logits are as close to that evaluation as it is to its own re-ordered twin (margin: half the
    #     north-star tolerance in max-rel, a fifth of it in rel-L2)
    assert table[allk]["vs_hip"] <= floor["maxrel"] + 5e-4, table
    assert table[allk]["relL2_vs_hip"] <= floor["relL2"] + 2e-4, table
    # (b) moving the rounding points to the HIP path's places explains the gap to the reference-fp16 logits ...
    assert table[allk]["vs_hip"] < table["reference rounding"]["vs_hip"], table
    # (c) ... and does not cost accuracy: against the fp32 truth the HIP path is no worse than the reference's own rounding
    assert table["hip"]["vs_fp32_truth"] <= table["reference rounding"]["vs_fp32_truth"] + 1e-3, table


def test_full_size_fp32_residual_stream_question():
    """VERDICT r5 #6b, answered with numbers (oracle only; nothing here touches the HIP path): can ANY design with fp16 GEMM operands meet the
    literal north-star tolerance -- MLM logits within 1e-3 max-rel of the fp32 truth -- at B = 64, L = 167, 12 layers?  The oracle is evaluated
    in fp16 at the HIP path's rounding points, then with the residual stream kept in fp32 (`residual_fp32`: one extra fp32 stream per LayerNorm,
    the pre-LayerNorm sum still rounded to fp16) and with the pre-LayerNorm sums in fp32 as well (`residual_fp32_full`: fp32 outputs on both
    residual GEMMs of every layer).  The table goes to gpurun_out/parity_residual_fp32.json -> profiles/; DESIGN.md section 4 quotes it.
    Asserted: only the ordering (each step helps) and that the cheap variant does NOT reach 1e-3 -- which is why it is not offered as a switch."""
    c = FULL_CASES["coco_s2s"]
    p = O.init_params(vocab_size=V, layers=12, tasks="img2txt", seed=c["seed"])
    batch = S.make_batch(B, max_len_b=64, vocab_size=V, max_pred=3, s2s_prob=1.0, tasks="img2txt", seed=c["seed"] + 7)
    truth, _ = _oracle(p, batch, "img2txt", torch.float32)
    t = truth["mlm_logits"].float()
    # the same fp32 evaluation on the fp16-ROUNDED weights: how much of every fp16 row below is operand (weight) rounding alone
    p16 = {k: v.half().float() for k, v in p.items()}
    tw, _ = _oracle(p16, batch, "img2txt", torch.float32)
    table = {"fp32_arithmetic_on_fp16_rounded_weights": {"maxrel_vs_fp32_truth": _relmax(tw["mlm_logits"].float(), t), "relL2": _relL2(tw["mlm_logits"].float(), t)}}
    for name, flags in (("reference_rounding", ()), ("hip_rounding_points", O.ROUNDING_FLAGS),
                        ("hip_rounding_points+residual_fp32", O.ROUNDING_FLAGS + ("residual_fp32",)),
                        ("hip_rounding_points+residual_fp32_full", O.ROUNDING_FLAGS + ("residual_fp32_full",))):
        with O.rounding(*flags):
            out, _ = _oracle(p, batch, "img2txt", torch.float16)
        l = out["mlm_logits"].float().reshape(t.shape)
        table[name] = {"maxrel_vs_fp32_truth": _relmax(l, t), "relL2": _relL2(l, t)}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_residual_fp32.json", "w") as f:
        json.dump(table, f, indent=1)
    e = {k: v["maxrel_vs_fp32_truth"] for k, v in table.items()}
    assert e["hip_rounding_points+residual_fp32_full"] <= e["hip_rounding_points+residual_fp32"] <= e["hip_rounding_points"] + 1e-4, table
    assert e["hip_rounding_points+residual_fp32"] > 1e-3, table          # the cheap variant does not buy the literal tolerance


def test_full_size_logits_under_every_nt_variant():
    """Every vlp_gemm_nt variant the table / an autotune run may select is forced for ALL forward GEMMs at M = 10 688 (tile tails
    under each tile shape and XCD remap), logits and loss checked against the fp32 oracle."""
    from vlp_amd.engine import Engine
    p = O.init_params(vocab_size=V, layers=12, tasks="img2txt", seed=111)
    batch = S.make_batch(B, max_len_b=64, vocab_size=V, max_pred=3, s2s_prob=0.75, seed=112)
    truth, _ = _oracle(p, batch, "img2txt", torch.float32)
    ref16, _ = _oracle(p, batch, "img2txt", torch.float16)
    t = truth["mlm_logits"].float()
    yard = _relmax(ref16["mlm_logits"].float(), t)
    m = _build(p, "img2txt")
    from vlp_amd import _lib as K
    old = Engine.GEMM_NT_VARIANT
    rep, ran = {}, {}
    try:
        for v in sorted(set(Engine.NT_CANDIDATES) | {1, 3, 11, 13}):
            Engine.GEMM_NT_VARIANT = v
            losses = _hip(m, batch)
            # the last GEMM of the forward is the tied decoder (plain epilogue + bias): the launcher must have run the forced variant
            # there (vlp_gemm_nt_resolved_variant: what ran after the launcher's fallbacks)
            ran[v] = K.gemm_nt_resolved_variant()
            # (the persistent k-stream kernel carries N % 128 == 0 only: the 28 996-column decoder must report its ring fallback; the
            # encoder's QKV / FFN GEMMs of this forward did run on it: tests/test_00 asserts the resolved variant per epilogue)
            assert ran[v] == (29 if v & 256 else v), (v, ran[v])
            rep[v] = _relmax(m.last_mlm_logits.float().reshape(t.shape), t)
            assert rep[v] <= yard + 1e-3, (v, rep, yard)
            assert abs(float(losses[0]) - float(truth["mlm_loss"])) <= 2e-3 * float(truth["mlm_loss"]), v
    finally:
        Engine.GEMM_NT_VARIANT = old
    # (the numbers may coincide to the last digit: variants of one MFMA family accumulate every output element in the same order whatever
    # the tile shape, and tests/test_00_kernels_gpu.py::test_gemm_nt_variant_identity records whether the two families agree bit for bit)
    FULL_REPORT["nt_variants_logits_vs_fp32"] = {"reference_fp16": yard, **{str(k): x for k, x in rep.items()}}
    FULL_REPORT["nt_variants_resolved"] = {str(k): x for k, x in ran.items()}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_fullsize.json", "w") as f:
        json.dump(FULL_REPORT, f, indent=1)
