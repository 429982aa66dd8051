"""Host enqueue time per training step vs GPU time (same setup as bench.py)."""
import sys, time, torch
sys.path.insert(0, ".")
from vlp_amd import synthetic as S
from vlp_amd.modeling import BertConfig, BertForPreTrainingLossMask
from vlp_amd.optimization_fp16 import FP16_Optimizer_State, FusedAdam
from vlp_amd.run_img2txt_dist import train_step
dev = torch.device("cuda:0")
cfg = BertConfig(28996, num_hidden_layers=12, type_vocab_size=6)
model = BertForPreTrainingLossMask(cfg, enable_butd=True, len_vis_input=100, allow_random_fc7=True).half().to(dev).train()
named = list(model.named_parameters()); nd = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
groups = [{"params": [p for n, p in named if not any(x in n for x in nd)], "weight_decay": 0.01}, {"params": [p for n, p in named if any(x in n for x in nd)], "weight_decay": 0.0}]
opt = FP16_Optimizer_State(FusedAdam(groups, lr=3e-5, bias_correction=False, max_grad_norm=1.0), dynamic_loss_scale=True)
b = S.batch_to(S.make_batch(64, max_len_b=64, vocab_size=28996, seed=1), dev, half=True)
for _ in range(4): train_step(model, opt, b, 1e-5)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 5
for _ in range(n): train_step(model, opt, b, 1e-5)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): train_step(model, opt, b, 1e-5)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
