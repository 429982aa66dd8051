"""CPU: the oracle restatement reproduces the fixtures that the UNMODIFIED reference produced
(tests/golden/*.npz, generator: oracle/make_golden.py).  Runs anywhere (no /root/reference)."""
import numpy as np
import pytest
import torch

from oracle import vlp_oracle as O
from tests.golden_util import CASES, load_case, rel_err, sample

TOL = 2e-5   # fp32 CPU vs fp32 CPU, different op order


@pytest.mark.parametrize("name", [n for n in CASES if "12l" not in n] + ["img2txt_L167_12l"])
def test_oracle_matches_reference_fixture(name):
    try:
        g, p, batch, mk = load_case(name)
    except RuntimeError as e:
        pytest.skip(str(e))
    p = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    out, grads = O.loss_and_grads(p, batch, tasks=mk["tasks"], capture=True, mask_image_regions=mk.get("mask_image_regions", False))
    if "pooled_output" in g:          # vismask cases: the pooler is live (modeling.py:411-417, 1124)
        assert rel_err(out["pooled_output"].detach().numpy(), g["pooled_output"]) < TOL
        assert g["losses"][1] > 0.5
    losses = [float(out[k].sum()) for k in ("mlm_loss", "vis_pretext_loss", "vqa_loss")]
    assert np.allclose(losses, g["losses"], rtol=1e-5, atol=1e-6)
    # shapes of the 3-tuple: the live loss is 0-dim, the placeholders are [1]
    assert [out[k].dim() for k in ("mlm_loss", "vis_pretext_loss", "vqa_loss")] == list(g["loss_shapes"])
    if "mlm_logits" in g:
        assert rel_err(out["mlm_logits"].detach().numpy(), g["mlm_logits"]) < TOL
    if "vqa_logits" in g:
        assert rel_err(out["vqa_logits"].detach().numpy(), g["vqa_logits"]) < TOL
    hid = [out["emb"]] + out["hidden"]
    for i, h in enumerate(hid):
        assert rel_err(sample(h), g["hidden_%d" % i]) < 5e-5, i
    gscale = max(x for x in g["grad_norms"] if x > 0)
    for n, ref_norm in zip(g["param_names"], g["grad_norms"]):
        gr = grads[str(n)]
        if ref_norm < 0:          # reference: .grad is None (unused parameter)
            assert gr is None or float(gr.abs().max()) == 0.0, n
        else:
            assert abs(float(gr.double().norm()) - ref_norm) <= 1e-4 * ref_norm + 1e-7 * gscale, n
    for k in g:
        if k.startswith("grad::"):
            n = k[6:]
            assert np.abs(sample(grads[n]) - g[k]).max() <= 1e-4 * np.abs(g[k]).max() + 1e-9, n


def test_bert_adam_restatement_matches_fixture():
    name = "img2txt_L123_2l"
    try:
        g, p, batch, mk = load_case(name)
    except RuntimeError as e:
        pytest.skip(str(e))
    p0 = {k: v.clone() for k, v in p.items()}
    p = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    _, grads = O.loss_and_grads(p, batch, tasks=mk["tasks"])
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    for k in g:
        if not k.startswith("adam::"):
            continue
        n = k[6:]
        w = p0[n].clone()
        m, v = torch.zeros_like(w), torch.zeros_like(w)
        step = 0
        for _ in range(2):
            step = O.bert_adam_step(w, grads[n], m, v, step, lr=1e-2, warmup=0.1, t_total=20,
                                    weight_decay=0.0 if any(nd in n for nd in no_decay) else 0.01)
        upd = sample(w - p0[n])
        assert np.abs(upd - g[k]).max() <= 1e-4 * np.abs(g[k]).max() + 2e-8, n


def test_schedules():
    assert O.warmup_linear(0.05, 0.1) == pytest.approx(0.5)
    assert O.warmup_linear(0.1, 0.1) == pytest.approx(1.0)
    assert O.warmup_linear(0.55, 0.1) == pytest.approx(0.5)
    assert O.warmup_linear(1.5, 0.1) == 0
    assert O.warmup_constant(0.5, 0.1) == 1.0
