"""Size-independent properties at BASELINE.json's FULL size (BERT-base 12 layers, vocab 28 996, 100 regions, seq 64 -> L = 167, batch 64),
where the oracle is too slow to be the checker: bitwise determinism, exact equivariance under a permutation of the samples, the
expected loss of an untrained model (ln V), and linearity of the backward pass in the loss scale."""
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
