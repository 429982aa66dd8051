"""GPU: optimizer parity (fused Adam / BertAdam vs the oracle's restatements), the train-loop contract, the entry
script on synthetic data, checkpoint round trip."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from oracle import vlp_oracle as O                                      # noqa: E402 (checker)
from vlp_amd import synthetic as S                                      # noqa: E402
from vlp_amd.modeling import BertConfig, BertForPreTrainingLossMask     # noqa: E402
from vlp_amd.optimization import BertAdam, warmup_linear                # noqa: E402
from vlp_amd.optimization_fp16 import FP16_Optimizer_State, FusedAdam   # noqa: E402
from vlp_amd.run_img2txt_dist import train_step                         # noqa: E402

DEV = torch.device("cuda:0")
ND = ["bias", "LayerNorm.bias", "LayerNorm.weight"]


def small_model(tasks="img2txt", drop=0.0, seed=3, layers=2):
    p = O.init_params(vocab_size=1024, layers=layers, tasks=tasks, seed=seed)
    cfg = BertConfig(1024, num_hidden_layers=layers, type_vocab_size=6, hidden_dropout_prob=drop, attention_probs_dropout_prob=drop)
    m = BertForPreTrainingLossMask(cfg, enable_butd=True, len_vis_input=100, tasks=tasks, allow_random_fc7=True)
    sd = dict(p)
    sd["cls.predictions.decoder.weight"] = p["bert.embeddings.word_embeddings.weight"]
    m.load_state_dict(sd, strict=True)
    return m.half().to(DEV), p


def groups_of(model):
    named = list(model.named_parameters())
    return [{"params": [q for n, q in named if not any(x in n for x in ND)], "weight_decay": 0.01},
            {"params": [q for n, q in named if any(x in n for x in ND)], "weight_decay": 0.0}]


def fwd_bwd(model, opt, batch):
    b = batch
    lt = model(b.img, b.vis_pe, b.input_ids, b.segment_ids, b.input_mask, b.lm_label_ids, b.ans_labels, b.is_next, masked_pos=b.masked_pos,
               masked_weights=b.masked_weights, task_idx=b.task_idx, vis_masked_pos=b.vis_masked_pos, drop_worst_ratio=0)
    opt.backward(lt[0] + lt[1] + lt[2])
    return lt


def test_fp16_optimizer_step_matches_apex_restatement():
    model, _ = small_model()
    model.train()
    opt = FP16_Optimizer_State(FusedAdam(groups_of(model), lr=1e-3, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True)
    batch = S.batch_to(S.make_batch(4, max_len_b=20, vocab_size=1024, max_pred=3, seed=1), DEV, half=True)
    eng = model.engine
    assert opt.cur_scale == 65536.0
    for it in range(2):
        fwd_bwd(model, opt, batch)
        want = []
        for i, key in enumerate(opt._group_key):
            g16 = eng.gflat[key].clone()
            norm = float(g16.float().norm())
            p32, m, v = opt.fp32_groups_flat[i].clone(), opt._m[i].clone(), opt._v[i].clone()
            O.fused_adam_step(p32, g16, m, v, lr=1e-3, grad_norm_scaled=norm, scale=opt.cur_scale, weight_decay=opt.param_groups[i]["weight_decay"])
            want.append((p32, m, v))
        for g in opt.param_groups:
            g["lr"] = 1e-3
        opt.step()
        opt.zero_grad()
        assert not opt.overflow
        for i, key in enumerate(opt._group_key):
            p32, m, v = want[i]
            assert float((opt.fp32_groups_flat[i] - p32).abs().max()) <= 1e-6 * float(p32.abs().max()) + 1e-9
            assert float((opt._m[i] - m).abs().max()) <= 1e-5 * float(m.abs().max()) + 1e-12
            assert float((eng.flat[key].float() - p32).abs().max()) <= 1e-3 * float(p32.abs().max())    # fp16 model copy
    assert opt.cur_iter == 2


def test_ten_step_trajectory_against_the_independent_fp32_oracle():
    """The HIP path (fp16 storage, FP16_Optimizer + FusedAdam, dynamic loss scale) and the oracle (fp32 forward / autograd backward /
    fused_adam_step on ITS OWN gradients) start from the same weights and see the same ten batches: the loss of every step and the
    final weights must agree to fp16 working precision.  Unlike the one-step optimizer tests nothing of the HIP path feeds the oracle."""
    LR, STEPS = 2e-4, 10
    model, p0 = small_model(drop=0.0)
    model.train()
    # a start scale the first steps do not overflow at: a skipped step is the scaler's business (tested above), not this comparison's
    opt = FP16_Optimizer_State(FusedAdam(groups_of(model), lr=LR, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True,
                               dynamic_loss_args={"init_scale": 2.0 ** 10})
    ref = {k: v.clone().to(DEV).requires_grad_(True) for k, v in p0.items()}
    rm = {k: torch.zeros_like(v) for k, v in ref.items()}
    rv = {k: torch.zeros_like(v) for k, v in ref.items()}
    is_nd = {k: any(x in k for x in ND) for k in ref}
    hip_losses, ref_losses = [], []
    for it in range(STEPS):
        raw = S.make_batch(4, max_len_b=20, vocab_size=1024, max_pred=3, seed=100 + it)
        lt = train_step(model, opt, S.batch_to(raw, DEV, half=True), LR)
        hip_losses.append(float(lt[0].detach()))
        out, grads = O.loss_and_grads(ref, S.batch_to(raw, DEV), tasks="img2txt")
        ref_losses.append(float(out["loss"].sum().detach()))
        used = [k for k in ref if grads[k] is not None]
        norm = {nd: float(torch.sqrt(sum((grads[k].double() ** 2).sum() for k in used if is_nd[k] == nd))) for nd in (False, True)}
        with torch.no_grad():
            for k in used:      # apex: one clip factor per param group (fp16_optimizer norm_groups), scale 1 on the fp32 side
                O.fused_adam_step(ref[k], grads[k], rm[k], rv[k], lr=LR, grad_norm_scaled=norm[is_nd[k]], scale=1.0,
                                  weight_decay=0.0 if is_nd[k] else 0.01)
        for k in ref:
            ref[k].grad = None
    assert not opt.overflow and opt.skipped_steps == 0
    rel = [abs(a - b) / abs(b) for a, b in zip(hip_losses, ref_losses)]
    print("trajectory: max rel loss difference %.2e" % max(rel), hip_losses, ref_losses)
    assert max(rel) <= 5e-4, (hip_losses, ref_losses)              # measured 9.1e-5 over the ten steps (fp16 logits under a 7-nat loss)
    # weights: distance between the two end points relative to the distance travelled (Adam moves every weight by ~LR per step whatever
    # the size of its gradient, so where a gradient is at the fp16 noise level the two runs may step in different directions)
    d2 = u2 = 0.0
    worst = 0.0
    for i, key in enumerate(opt._group_key):
        flat = opt.fp32_groups_flat[i]
        offs = model.engine.offsets[key]
        for n, q in model.named_parameters():
            if n in offs and float(rm[n].abs().max()) > 0:
                mine = flat[offs[n]:offs[n] + q.numel()].view_as(ref[n]).double()
                theirs, start = ref[n].detach().double(), p0[n].to(DEV).double()
                d2 += float(((mine - theirs) ** 2).sum())
                u2 += float(((theirs - start) ** 2).sum())
                worst = max(worst, float((mine - theirs).abs().max()))
    ratio = (d2 / u2) ** 0.5
    print("trajectory: |w_hip - w_oracle| / |w_oracle - w_0| = %.3f, worst element %.2e (%.1f LR)" % (ratio, worst, worst / LR))
    assert u2 > 0 and ratio <= 0.05, ratio                          # measured 0.016
    # no bias correction (the reference's FusedAdam configuration): a step moves a weight by up to (1 - b1) / sqrt(1 - b2) = 3.16 LR
    assert worst <= 2 * STEPS * LR * 3.17, worst

def test_overflow_skips_step_and_halves_scale():
    model, _ = small_model()
    model.train()
    opt = FP16_Optimizer_State(FusedAdam(groups_of(model), lr=1e-3, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True,
                               dynamic_loss_args={"init_scale": 2.0 ** 30})
    batch = S.batch_to(S.make_batch(4, max_len_b=20, vocab_size=1024, max_pred=3, seed=1), DEV, half=True)
    before = model.engine.flat["decay"].clone()
    fwd_bwd(model, opt, batch)                 # 2^30 x gradient overflows fp16
    opt.step()
    opt.zero_grad()
    assert opt.overflow and opt.cur_scale == 2.0 ** 29 and opt.skipped_steps == 1
    assert torch.equal(before, model.engine.flat["decay"])


def test_training_reduces_loss_and_checkpoint_round_trip(tmp_path):
    model, _ = small_model(drop=0.1)
    model.train()
    opt = FP16_Optimizer_State(FusedAdam(groups_of(model), lr=2e-4, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True)
    batch = S.batch_to(S.make_batch(8, max_len_b=20, vocab_size=1024, max_pred=3, seed=2), DEV, half=True)
    losses = []
    for it in range(30):
        lt = train_step(model, opt, batch, 2e-4)
        losses.append(float(lt[0]))
    assert all(l == l for l in losses)                       # no NaN
    assert sum(losses[-5:]) / 5 < 0.7 * sum(losses[:5]) / 5, losses
    # checkpoint: same keys as the reference, loads into a fresh model and reproduces the eval loss
    import copy
    sd = copy.deepcopy(model).cpu().state_dict()
    assert "cls.predictions.decoder.weight" in sd and all(not v.is_cuda for v in sd.values())
    torch.save(sd, os.path.join(tmp_path, "model.1.bin"))
    m2, _ = small_model(seed=99)
    m2.load_state_dict(torch.load(os.path.join(tmp_path, "model.1.bin")), strict=True)
    m2 = m2.half().to(DEV).eval()
    model.eval()
    b = batch
    args = (b.img, b.vis_pe, b.input_ids, b.segment_ids, b.input_mask, b.lm_label_ids, b.ans_labels, b.is_next)
    kw = dict(masked_pos=b.masked_pos, masked_weights=b.masked_weights, drop_worst_ratio=0)
    with torch.no_grad():
        assert float(model(*args, **kw)[0]) == float(m2(*args, **kw)[0])
    # optimizer state_dict carries the reference's fields (optimization_fp16.py:28-37)
    osd = opt.state_dict()
    for k in ("dynamic_loss_scale", "cur_scale", "cur_iter", "last_overflow_iter", "scale_factor", "scale_window", "optimizer_state_dict",
              "fp32_groups_flat"):
        assert k in osd
    opt.load_state_dict(osd)


def test_training_with_masked_regions_learns_the_pretext_task():
    """--vis_mask_prob 0.25 through the train loop (run_img2txt_dist.py:194,482,531): the summed loss (MLM + vis_pretext) falls on a
    repeated batch, the pretext loss itself falls (its gradient reaches the region projections and the pooler), the pooler's
    parameters move (they are frozen without the branch), dropout 0.1 active, bitwise reproducible."""
    def run():
        model, _ = small_model(drop=0.1)
        model.train()
        opt = FP16_Optimizer_State(FusedAdam(groups_of(model), lr=2e-4, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True)
        batch = S.batch_to(S.make_batch(8, max_len_b=20, vocab_size=1024, max_pred=3, seed=2, vis_mask_prob=0.25), DEV, half=True)
        w0 = model.bert.pooler.dense.weight.detach().float().clone()
        hist = []
        for it in range(30):
            lt = train_step(model, opt, batch, 2e-4, mask_image_regions=True)
            hist.append((float(lt[0]), float(lt[1])))
        assert lt[1].dim() == 0 and lt[0].dim() == 0 and tuple(lt[2].shape) == (1,)              # the reference's tuple shapes
        return hist, w0, model
    hist, w0, model = run()
    assert all(a == a and b == b for a, b in hist)
    assert sum(b for _, b in hist[-5:]) < 0.8 * sum(b for _, b in hist[:5]), hist                  # the pretext task is being learned
    assert sum(a for a, _ in hist[-5:]) < 0.8 * sum(a for a, _ in hist[:5]), hist
    assert float((model.bert.pooler.dense.weight.detach().float() - w0).abs().max()) > 0
    assert "bert.pooler.dense.weight" not in model.engine.unused_parameter_names()
    hist2, _, _ = run()
    assert hist == hist2


def test_pipelined_optimizer_step_is_bit_identical():
    """FP16_Optimizer_State.pipeline_with_forward: the update runs on the optimizer stream, chunked in the order the next forward reads
    the parameters, and the forward waits per chunk.  Same kernels, same arithmetic: losses of every step, parameters, Adam moments
    and the loss-scale state must equal the plain step() bit for bit -- including an overflow (skipped) step."""
    runs = []
    for pipelined in (False, True):
        model, _ = small_model(drop=0.1, layers=3)
        model.train()
        opt = FP16_Optimizer_State(FusedAdam(groups_of(model), lr=3e-4, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True,
                                   dynamic_loss_args={"init_scale": 2.0 ** 24})          # the first steps overflow and are skipped
        opt.pipeline_with_forward = pipelined
        batches = [S.batch_to(S.make_batch(6, max_len_b=20, vocab_size=1024, max_pred=3, seed=40 + i), DEV, half=True) for i in range(3)]
        losses = []
        for it in range(12):
            if pipelined and it % 2 == 1:
                # hold the optimizer stream back: every parameter / loss-scale read of the next forward and backward must be
                # ORDERED behind the step, not merely later in wall time -- on this 3-layer model the update otherwise finishes long
                # before anything reads it.  ~20 ms of spinning in front of the step's kernels.
                with torch.cuda.stream(model.engine.optimizer_stream()):
                    torch.cuda._sleep(40_000_000)
            lt = train_step(model, opt, batches[it % 3], 3e-4)
            losses.append(lt[0].detach().clone())
        torch.cuda.synchronize()
        runs.append((losses, {k: v.clone() for k, v in model.engine.flat.items()}, [t.clone() for t in opt.fp32_groups_flat],
                     [t.clone() for t in opt._m], [t.clone() for t in opt._v], opt.cur_scale, opt.skipped_steps, opt.applied_steps))
    a, b = runs
    assert b[6] >= 1 and a[5:] == b[5:], (a[5:], b[5:])
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k
    for i in (2, 3, 4):
        for x, y in zip(a[i], b[i]):
            assert torch.equal(x, y)


def test_apex_layout_optimizer_checkpoint_round_trip(tmp_path):
    """Checkpoint interchange with the reference stack (optimization_fp16.py:17-80 on apex FP16_Optimizer + FusedAdam): apex_state_dict()
    emits `optimizer_state_dict = {state: {gid: {step, exp_avg, exp_avg_sq}}, param_groups: [{..., params: [gid]}]}` + dense
    `fp32_groups_flat` in the caller's parameter order; a torch Optimizer built the way apex builds its inner optimizer (one flat fp32
    parameter per group) loads it; load_state_dict() of a fresh vlp_amd optimizer recognises the layout and continues bit for bit."""
    model, p0 = small_model(drop=0.0)
    model.train()
    opt = FP16_Optimizer_State(FusedAdam(groups_of(model), lr=3e-4, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True)
    batches = [S.batch_to(S.make_batch(4, max_len_b=20, vocab_size=1024, max_pred=3, seed=60 + i), DEV, half=True) for i in range(2)]
    for it in range(3):
        train_step(model, opt, batches[it % 2], 3e-4)
    sd = opt.apex_state_dict()
    inner = sd["optimizer_state_dict"]
    assert set(inner) == {"state", "param_groups"} and [g["params"] for g in inner["param_groups"]] == [[0], [1]]
    assert all(set(inner["state"][i]) == {"step", "exp_avg", "exp_avg_sq"} and inner["state"][i]["step"] == 3 for i in (0, 1))
    # dense, caller's order: the flat master of a group == its parameters concatenated in named_parameters order
    for i, g in enumerate(groups_of(model)):
        want = torch.cat([q.detach().reshape(-1) for q in g["params"]])
        assert sd["fp32_groups_flat"][i].numel() == want.numel() == inner["state"][i]["exp_avg"].numel()
        assert torch.equal(sd["fp32_groups_flat"][i].half(), want)
    # what apex's inner optimizer looks like to torch: one flat fp32 parameter per group
    flats = [torch.nn.Parameter(t.clone()) for t in sd["fp32_groups_flat"]]
    ref_inner = torch.optim.Adam([{"params": [flats[0]], "weight_decay": 0.01}, {"params": [flats[1]], "weight_decay": 0.0}], lr=3e-4)
    ref_inner.load_state_dict({"state": {k: {kk: (torch.tensor(float(vv)) if kk == "step" else vv) for kk, vv in v.items()} for k, v in inner["state"].items()},
                               "param_groups": [dict(g, betas=tuple(g["betas"]), amsgrad=False, maximize=False, foreach=None, capturable=False,
                                                     differentiable=False, fused=None) for g in inner["param_groups"]]})
    assert torch.equal(ref_inner.state[flats[0]]["exp_avg"], inner["state"][0]["exp_avg"])
    path = os.path.join(tmp_path, "optim.1.bin")
    torch.save({k: (v if not torch.is_tensor(v) else v.cpu()) for k, v in sd.items()}, path)
    # resume in a fresh model + optimizer from the apex-layout file
    m2, _ = small_model(drop=0.0)
    m2.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
    m2 = m2.half().to(DEV)
    m2.train()
    opt2 = FP16_Optimizer_State(FusedAdam(groups_of(m2), lr=3e-4, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True)
    opt2.load_state_dict(torch.load(path, map_location=DEV))
    for a, b in zip(opt._m + opt._v + opt.fp32_groups_flat, opt2._m + opt2._v + opt2.fp32_groups_flat):
        assert torch.equal(a, b)
    assert opt2.cur_scale == opt.cur_scale and opt2.cur_iter == opt.cur_iter and opt2.applied_steps == opt.applied_steps
    la = train_step(model, opt, batches[1], 3e-4)
    lb = train_step(m2, opt2, batches[1], 3e-4)
    torch.cuda.synchronize()
    assert torch.equal(la[0], lb[0])
    for k in model.engine.flat:
        assert torch.equal(model.engine.flat[k], m2.engine.flat[k]), k


def test_exact_resume_from_model_and_optimizer_checkpoint(tmp_path):
    """N4: the optimizer checkpoint the reference left disabled (run_img2txt_dist.py:599) and the resume path (:310,:428-437):
    2 epochs in one go == 1 epoch, process "restart", resume for epoch 2 -- bit for bit, with dropout on (the engine's mask-stream
    position travels with the optimizer state) and the dynamic loss scaler's state."""
    from vlp_amd import run_img2txt_dist as R
    common = ["--bert_model", "bert-base-cased", "--from_scratch", "--fp16", "--enable_butd", "--len_vis_input", "100", "--new_segment_ids",
              "--synthetic", "3", "--num_train_epochs", "2", "--train_batch_size", "4", "--max_len_b", "20", "--num_hidden_layers", "2",
              "--learning_rate", "3e-4", "--warmup_proportion", "0.3", "--loss_scale", "0", "--seed", "7"]
    a, b = os.path.join(tmp_path, "a"), os.path.join(tmp_path, "b")
    R.main(common + ["--output_dir", a])
    R.main(common + ["--output_dir", b, "--stop_after_epoch", "1"])
    assert os.path.exists(os.path.join(b, "optim.1.bin")) and not os.path.exists(os.path.join(b, "model.2.bin"))
    R.main(common + ["--output_dir", b])                   # finds model.1.bin + optim.1.bin and continues with epoch 2
    sa, sb = torch.load(os.path.join(a, "model.2.bin")), torch.load(os.path.join(b, "model.2.bin"))
    assert sa.keys() == sb.keys()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    oa, ob = torch.load(os.path.join(a, "optim.2.bin")), torch.load(os.path.join(b, "optim.2.bin"))
    assert oa["cur_iter"] == ob["cur_iter"] == 6 and oa["cur_scale"] == ob["cur_scale"]
    for x, y in zip(oa["fp32_groups_flat"] + oa["optimizer_state_dict"]["exp_avg"] + oa["optimizer_state_dict"]["exp_avg_sq"],
                    ob["fp32_groups_flat"] + ob["optimizer_state_dict"]["exp_avg"] + ob["optimizer_state_dict"]["exp_avg_sq"]):
        assert torch.equal(x, y) and not x.is_cuda
    # and the one-epoch model differs from the two-epoch one (the second epoch really ran)
    assert not torch.equal(torch.load(os.path.join(b, "model.1.bin"))["bert.encoder.layer.0.output.dense.weight"],
                           sb["bert.encoder.layer.0.output.dense.weight"])


def test_bert_adam_on_model_matches_reference_restatement():
    model, p0 = small_model()
    model.train()
    opt = BertAdam(groups_of(model), lr=1e-3, warmup=0.1, t_total=20)
    batch = S.batch_to(S.make_batch(4, max_len_b=20, vocab_size=1024, max_pred=3, seed=1), DEV, half=True)
    b = batch
    ref = {k: v.clone() for k, v in p0.items()}
    rm = {k: torch.zeros_like(v) for k, v in ref.items()}
    rv = {k: torch.zeros_like(v) for k, v in ref.items()}
    unused = model.engine.unused_parameter_names() if model.engine.packed else {"bert.pooler.dense.weight", "bert.pooler.dense.bias"}
    for step in range(2):
        lt = model(b.img, b.vis_pe, b.input_ids, b.segment_ids, b.input_mask, b.lm_label_ids, b.ans_labels, b.is_next, masked_pos=b.masked_pos,
                   masked_weights=b.masked_weights, drop_worst_ratio=0)
        (lt[0] + lt[1] + lt[2]).backward()
        grads = {n: q.grad.detach().float().cpu().clone() for n, q in model.named_parameters()}
        for n in ref:
            if n in unused:
                continue
            O.bert_adam_step(ref[n], grads[n], rm[n], rv[n], step, lr=1e-3, warmup=0.1, t_total=20,
                             weight_decay=0.0 if any(x in n for x in ND) else 0.01)
        opt.step()
        opt.zero_grad()
    for n, q in model.named_parameters():
        # fp32 masters inside the optimizer follow the restatement; the model holds their fp16 rounding
        assert float((q.detach().float().cpu() - ref[n]).abs().max()) <= 1.5e-3 * float(ref[n].abs().max()) + 1e-6, n
    assert opt.get_lr()[0] == pytest.approx(1e-3 * warmup_linear(2 / 20, 0.1))


def test_entry_script_synthetic(tmp_path):
    from vlp_amd import run_img2txt_dist as R
    out = os.path.join(tmp_path, "run")
    R.main(["--output_dir", out, "--do_train", "--fp16", "--enable_butd", "--new_segment_ids", "--from_scratch", "--max_len_b", "20",
            "--train_batch_size", "4", "--num_train_epochs", "1", "--synthetic", "3", "--num_hidden_layers", "2", "--len_vis_input", "100"])
    assert os.path.exists(os.path.join(out, "model.1.bin")) and os.path.exists(os.path.join(out, "opt.json"))
    sd = torch.load(os.path.join(out, "model.1.bin"))
    assert "bert.encoder.layer.1.output.LayerNorm.bias" in sd and sd["bert.embeddings.word_embeddings.weight"].shape == (28996, 768)
    # --vis_mask_prob > 0 (mask_image_regions + vis_pretext_loss): the same entry point, pooler trained
    out2 = os.path.join(tmp_path, "run_vm")
    R.main(["--output_dir", out2, "--do_train", "--fp16", "--enable_butd", "--new_segment_ids", "--from_scratch", "--max_len_b", "20",
            "--train_batch_size", "4", "--num_train_epochs", "1", "--synthetic", "3", "--num_hidden_layers", "2", "--len_vis_input", "100",
            "--vis_mask_prob", "0.25"])
    assert os.path.exists(os.path.join(out2, "model.1.bin"))


def test_entry_script_reference_readme_command_line(tmp_path):
    """The reference README's example commands pass --amp WITHOUT --fp16 (README.md:112,131): in the reference that is the fp32 BertAdam path
    (amp engages only with --fp16, run_img2txt_dist.py:305; optimizer :421-426).  Here the command line stops with a message that says so and
    names both ways on; with --allow_fp16_compute it trains under BertAdam (own warm-up schedule, fp32 master weights, fp16 compute, static
    loss scale on the backward): finite weights that moved, a loss that falls on a repeated synthetic pool."""
    from vlp_amd import run_img2txt_dist as R
    base = ["--do_train", "--new_segment_ids", "--always_truncate_tail", "--amp", "--enable_butd", "--s2s_prob", "1", "--bi_prob", "0", "--from_scratch",
            "--max_len_b", "20", "--train_batch_size", "4", "--num_hidden_layers", "2", "--len_vis_input", "100", "--learning_rate", "1e-4", "--log_every", "1"]
    with pytest.raises(NotImplementedError) as e:
        R.main(["--output_dir", os.path.join(tmp_path, "refuse"), "--num_train_epochs", "1", "--synthetic", "2"] + base)
    msg = str(e.value)
    assert "README" in msg and "--fp16" in msg and "--allow_fp16_compute" in msg and "BertAdam" in msg
    out = os.path.join(tmp_path, "run")
    R.main(["--output_dir", out, "--num_train_epochs", "3", "--synthetic", "4", "--allow_fp16_compute"] + base)
    sd = torch.load(os.path.join(out, "model.3.bin"))
    assert all(torch.isfinite(v.float()).all() for v in sd.values())
    osd = torch.load(os.path.join(out, "optim.3.bin"))
    assert "vlp_master_fp32" in osd and osd["param_groups"][0]["schedule"] == "warmup_linear"          # BertAdam's state, not FP16_Optimizer_State's
    losses = [float(l.rsplit("Loss", 1)[1]) for l in open(os.path.join(out, "training.log")) if "Iter" in l and "Loss" in l]
    assert len(losses) == 12 and all(x == x for x in losses) and sum(losses[-4:]) < sum(losses[:4]), losses          # the same 4 pooled batches, three passes


def test_entry_script_baseline_config0_shape(tmp_path):
    """BASELINE.json configs[0] through the entry script: 2 layers, 8 regions, seq_len 32, bs 4, world_size 1 (--local_rank -1), the reference's
    fp32 command line (no --fp16: BertAdam, run_img2txt_dist.py:421-426).  The reference runs it on the CPU; here the same command line plus
    --allow_fp16_compute runs it on the GPU under BertAdam with fp32 master weights (there is no CPU product path): checkpoints
    in the reference's key names, finite weights, a loss that falls over three passes of the same four pooled batches."""
    from vlp_amd import run_img2txt_dist as R
    out = os.path.join(tmp_path, "cfg0")
    R.main(["--output_dir", out, "--do_train", "--new_segment_ids", "--enable_butd", "--from_scratch", "--local_rank", "-1", "--len_vis_input", "8",
            "--max_len_b", "21", "--train_batch_size", "4", "--num_hidden_layers", "2", "--num_train_epochs", "3", "--synthetic", "4",
            "--learning_rate", "1e-4", "--log_every", "1", "--allow_fp16_compute"])
    sd = torch.load(os.path.join(out, "model.3.bin"))
    assert "bert.encoder.layer.1.output.LayerNorm.bias" in sd and "bert.encoder.layer.2.output.LayerNorm.bias" not in sd
    assert all(torch.isfinite(v.float()).all() for v in sd.values())
    osd = torch.load(os.path.join(out, "optim.3.bin"))
    assert "vlp_master_fp32" in osd
    losses = [float(l.rsplit("Loss", 1)[1]) for l in open(os.path.join(out, "training.log")) if "Iter" in l and "Loss" in l]
    assert len(losses) == 12 and all(x == x for x in losses) and sum(losses[-4:]) < sum(losses[:4]), losses


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag", "sharded"])
def test_bench_through_torchrun_and_rccl_world1(mode):
    """The launch line the driver uses for N > 1, with N = 1: RCCL process group, parameter broadcast, bucketed
    ReduceOp.AVG all-reduce hooks fired from the fused backward (mode rs_ag: reduce_scatter_tensor + all_gather_into_tensor on RCCL,
    so that branch is not first executed on the driver's 8-GPU node), barrier + max-over-ranks timing, rank checksum comparison."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", {"allreduce": "29517", "rs_ag": "29523", "sharded": "29525"}[mode], os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
           "--layers", "2", "--batch", "8", "--force-dist", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, VLP_DDP_MODE=mode))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["final_loss"] == out["config"]["final_loss"]
    assert out["config"]["rccl_ranks"] == 1 and out["config"]["rank_param_checksums_equal"] is True
    assert out["config"]["optimizer"] == ("sharded" if mode == "sharded" else "replicated")
    _check_comm_fields(out, mode)


def _check_comm_fields(out, mode):
    """bench.py's N > 1 line explains itself: collective form, bucket sizes, exposed / overlapped communication of the sampling pass."""
    c = out["config"]
    assert c["ddp_mode"] == mode and c["buckets"]["count"] == len(c["buckets"]["mb"]) and all(mb > 0 for mb in c["buckets"]["mb"])
    comm = c["comm"]
    assert comm is not None and comm["steps"] == out["steps"] and len(comm["per_bucket"]) == c["buckets"]["count"]
    assert comm["exposed_ms"] >= 0 and comm["overlapped_ms"] >= 0
    assert abs(comm["collectives_ms"] - sum(b["ms"] for b in comm["per_bucket"])) <= 1e-2 + 1e-3 * comm["collectives_ms"]
    # overlapped = per-step max(collectives - exposed, 0), averaged: between 0 and the collectives' own time (two ranks sharing one GPU over gloo
    # stall each other for tens of ms in single steps, so it is NOT max(mean collectives - mean exposed, 0))
    assert comm["overlapped_ms"] <= comm["collectives_ms"] + 1e-2
    assert [b["mb"] for b in comm["per_bucket"]] == c["buckets"]["mb"]
    if mode == "sharded":
        assert comm["param_gather_wait_ms_per_step"] >= 0


_WORLD2_CHECKSUMS = {}


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag", "sharded"])
def test_bench_world2_on_one_gpu_real_engine_ddp_hooks(mode):
    """The N > 1 path of bench.py / run_img2txt_dist.py with the REAL engine: two ranks (sharing this box's single GPU, backend gloo
    because RCCL refuses two ranks on one device) drive DistributedDataParallel -- parameter broadcast, grouped wgrads announcing
    their gradient bucket from the side stream, asynchronous bucket reductions, finish() before the optimizer -- on different data,
    and must hold bit-identical parameters afterwards (checked inside bench.py with VLP_BENCH_CHECK_RANKS=1)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", {"allreduce": "29519", "rs_ag": "29521", "sharded": "29527"}[mode], os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--layers", "3",
           "--batch", "16", "--no-cpu-baseline"]
    env = dict(os.environ, VLP_BENCH_SHARE_GPU="1", VLP_DIST_BACKEND="gloo", VLP_BENCH_CHECK_RANKS="1", VLP_DDP_MODE=mode)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and out["config"]["global_batch"] == 32
    assert out["value"] > 0 and out["config"]["final_loss"] == out["config"]["final_loss"]
    # the SHARDED optimizer step (each rank updates its half of master / m / v from its reduce-scattered gradient chunk, clip norm by a
    # 4-float all-reduce, parameters all-gathered) must leave the parameters the replicated rs_ag run leaves: the per-element update is
    # the same kernel on the same reduced gradient; only the global gradient norm is summed in another order (two partial sums), which
    # moves the clip factor by an ulp -- the three checksums agree to 1e-6 relative, and the ranks agree with each other bit for bit
    assert out["config"]["optimizer"] == ("sharded" if mode == "sharded" else "replicated")
    _check_comm_fields(out, mode)
    _WORLD2_CHECKSUMS[mode] = out["config"]["param_checksum"]
    if "rs_ag" in _WORLD2_CHECKSUMS and "sharded" in _WORLD2_CHECKSUMS:
        for a, b in zip(_WORLD2_CHECKSUMS["rs_ag"], _WORLD2_CHECKSUMS["sharded"]):
            assert abs(a - b) <= 1e-6 * abs(a) + 1e-3, _WORLD2_CHECKSUMS


def test_sharded_parameter_gathers_are_waited_for_before_the_weight_transposes():
    """after a sharded optimizer step the updated parameters arrive by asynchronous per-bucket all-gathers
    (Engine._param_works).  The next forward transposes EVERY weight matrix on the side stream (W^T shadows for the dgrad GEMMs): that
    stream must wait for all gathers first, or backward would use last step's weights for the chunks other ranks own.  RCCL refuses two
    ranks on one GPU, so the ordering is checked with stand-in work handles that record on which stream they were waited for."""
    model, _ = small_model(drop=0.0)
    model.train()
    eng = model.engine
    batch = S.batch_to(S.make_batch(4, max_len_b=20, vocab_size=1024, max_pred=3, seed=1), DEV, half=True)
    b = batch
    args = (b.img, b.vis_pe, b.input_ids, b.segment_ids, b.input_mask, b.lm_label_ids, b.ans_labels, b.is_next)
    kw = dict(masked_pos=b.masked_pos, masked_weights=b.masked_weights, task_idx=b.task_idx, vis_masked_pos=b.vis_masked_pos, drop_worst_ratio=0)
    model(*args, **kw)                                  # creates the side stream
    side, log = eng._side, []
    assert side is not None

    class Work(object):
        def __init__(self, name):
            self.name = name

        def wait(self):
            log.append(("wait", self.name, torch.cuda.current_stream() == side))

    class Plan(object):
        bucket_of_slice = {i: i for i in range(len(eng.buckets))}

    real = eng._refresh_shadows

    def refresh():
        log.append(("transpose", None, torch.cuda.current_stream() == side))
        real()
    eng._refresh_shadows = refresh
    eng.shard_plan = Plan()
    eng._param_works = dict([("nodecay", Work("nodecay"))] + [(i, Work(i)) for i in range(len(eng.buckets))])
    try:
        model(*args, **kw)
    finally:
        eng._refresh_shadows, eng.shard_plan, eng._param_works = real, None, None
    t = [i for i, e in enumerate(log) if e[0] == "transpose"]
    assert len(t) == 1 and log[t[0]][2], log            # one transpose launch, on the side stream
    waited_on_side_before = {e[1] for e in log[:t[0]] if e[0] == "wait" and e[2]}
    assert waited_on_side_before == {"nodecay"} | set(range(len(eng.buckets))), log
    # the main stream still waits per bucket where the forward first reads it
    assert {e[1] for e in log if e[0] == "wait" and not e[2]} == {"nodecay"} | set(range(len(eng.buckets))), log
