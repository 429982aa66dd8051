#!/usr/bin/env python
"""Ties bench.py's cpu_baseline (kind "port": the oracle restatement) to the reference's OWN code: in the build container, where
/root/reference exists, the UNMODIFIED reference (pytorch_pretrained_bert/modeling.py BertForPreTrainingLossMask +
optimization.py BertAdam, loaded by oracle/ref_loader.py) and the port run the same training step -- the loop body of
vlp/run_img2txt_dist.py:462-586 on a synthetic COCO-shape batch: forward, backward, BertAdam.step -- side by side on the same threads.
B = 16, L = 167 (100 regions + 64 tokens + 3), 12 layers, vocab 28 996, fp32 (BASELINE.md section 3 protocol).
Writes profiles/r06_cpu_baseline_reference_vs_port.json; bench.py quotes the ratio in cpu_baseline.sample.
usage: python tools/cpu_baseline_ref_vs_port.py [steps] [threads]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader, vlp_oracle as O      # noqa: E402
from vlp_amd import synthetic as S                   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
torch.set_num_threads(threads)
B, V = 16, 28996
batch = S.make_batch(B, max_len_b=64, vocab_size=V, max_pred=3, seed=1234)


def timed(step):
    step()                       # warm-up
    t0 = time.time()
    for _ in range(steps):
        step()
    return (time.time() - t0) / steps


# ---- the unmodified reference -------------------------------------------------------------------------------------------
ref = ref_loader.load_reference()
model = ref_loader.build_reference_model(dict(vocab_size=V, num_hidden_layers=12), tasks="img2txt", seed=0, drop_prob=0.1).train()
named = list(model.named_parameters())
no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]          # run_img2txt_dist.py:394-401
groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
          {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
opt = ref.optimization.BertAdam(groups, lr=3e-5, warmup=0.1, t_total=1000)


def ref_step():
    losses = model(batch.img, batch.vis_pe, batch.input_ids, batch.segment_ids, batch.input_mask, batch.lm_label_ids, batch.ans_labels,
                   batch.is_next, masked_pos=batch.masked_pos, masked_weights=batch.masked_weights, task_idx=batch.task_idx,
                   vis_masked_pos=batch.vis_masked_pos, mask_image_regions=False, drop_worst_ratio=0)
    loss = losses[0] + losses[1] + losses[2]                       # run_img2txt_dist.py:531
    loss.sum().backward()
    opt.step()
    opt.zero_grad()


t_ref = timed(ref_step)
del model, opt

# ---- the port (what bench.py times on the GPU box, where /root/reference does not exist) -----------------------------------
p = O.init_params(vocab_size=V, layers=12, tasks="img2txt", seed=0)
p = {k: v.requires_grad_(True) for k, v in p.items()}
m = {k: torch.zeros_like(v) for k, v in p.items()}
v2 = {k: torch.zeros_like(t) for k, t in p.items()}


def port_step():
    for t in p.values():
        t.grad = None
    O.loss_and_grads(p, batch, tasks="img2txt")
    with torch.no_grad():
        for k, t in p.items():
            if t.grad is not None:
                O.bert_adam_step(t, t.grad, m[k], v2[k], 1, lr=3e-5, warmup=0.1, t_total=1000,
                                 weight_decay=0.0 if ("bias" in k or "LayerNorm" in k) else 0.01)


t_port = timed(port_step)
out = {"what": "one training step (forward + backward + BertAdam.step), B=16, L=167, 12 layers, vocab 28996, fp32, CPU",
       "threads": threads, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(" :\t") if os.path.exists("/proc/cpuinfo") else "?",
       "steps_timed": steps, "reference_s_per_step": round(t_ref, 3), "port_s_per_step": round(t_port, 3),
       "reference_samples_per_s": round(B / t_ref, 3), "port_samples_per_s": round(B / t_port, 3),
       "reference_over_port": round(t_port / t_ref, 3),
       "note": "reference = /root/reference pytorch_pretrained_bert (unmodified, dropout 0.1, train mode) + optimization.BertAdam; port = oracle/vlp_oracle.py "
               "(dropout-free functional restatement + bert_adam_step): bench.py's cpu_baseline times the port on the GPU box's host cores"}
path = os.path.join(ROOT, "profiles", "r06_cpu_baseline_reference_vs_port.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
