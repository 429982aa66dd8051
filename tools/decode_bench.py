"""N1 measurement: captions/s of the K/V-cache decoder (12 layers, vocab 28 996, 100 regions, 20 generated tokens), greedy and beam."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vlp_amd import synthetic as S  # noqa: E402
from vlp_amd.modeling import BertConfig, BertForSeq2SeqDecoder  # noqa: E402

dev = torch.device("cuda:0")
B, T, Nv = int(os.environ.get("B", 64)), 20, 100
torch.manual_seed(0)
cfg = BertConfig(28996, num_hidden_layers=12, type_vocab_size=6)
out = {"batch": B, "new_tokens": T, "layers": 12}
g = torch.Generator().manual_seed(0)
img = torch.randn(B, Nv, 2048, generator=g).abs().half().to(dev)
vis_pe = torch.rand(B, Nv, 1607, generator=g).half().to(dev)
in_len, out_len = Nv + 2, Nv + 2 + T
input_ids = torch.tensor([[S.CLS_ID] + [S.UNK_ID] * Nv + [S.SEP_ID]] * B).to(dev)
token_type = torch.tensor([[4] * in_len + [5] * T] * B).to(dev)
pos = torch.arange(out_len).unsqueeze(0).expand(B, -1).contiguous().to(dev)
am = torch.zeros(B, out_len, out_len, dtype=torch.long)
am[:, :, :in_len] = 1
am[:, in_len:, in_len:] = torch.tril(torch.ones(T, T, dtype=torch.long))
am = am.to(dev)
MODES = [m for m in (("greedy", dict(search_beam_size=1)), ("beam3", dict(search_beam_size=3)), ("beam5", dict(search_beam_size=5))) if m[0] in os.environ.get("MODES", "greedy,beam3,beam5").split(",")]
for name, kw in MODES:
    m = BertForSeq2SeqDecoder(cfg, mask_word_id=S.MASK_ID, eos_id=S.SEP_ID, enable_butd=True, len_vis_input=Nv, allow_random_fc7=True, **kw).half().to(dev).eval()
    for _ in range(2):
        m(img, vis_pe, input_ids, token_type, pos, am)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        r = m(img, vis_pe, input_ids, token_type, pos, am)
    t_enq = (time.perf_counter() - t0) / n           # host time to enqueue (beam search also waits for its D2H copies here)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out[name] = {"ms_per_batch": round(dt * 1e3, 2), "captions_per_s": round(B / dt, 1), "ms_per_token_step": round(dt * 1e3 / T, 3),
                 "host_enqueue_ms": round(t_enq * 1e3, 2)}
    if name != "greedy":
        # beam_search() returns the best sequences, which needs one device -> host copy of the frames: the call itself waits for the GPU, so the
        # "enqueue" figure equals the wall time; the device work is enqueued in ~2 ms (Engine.decode_beam, hipGraph replays) -- decoding is not host-bound
        out[name]["host_enqueue_includes_final_device_to_host_copy"] = True
    if name == "greedy":
        # roofline of a TOKEN step (HBM-bound: every step streams the encoder + head weights and the K/V history once): algorithmic bytes =
        # fp16 weights of the 12 layers + head transform + tied vocabulary matrix, + K | V rows of every position decoded so far (B x Lk x 2H
        # fp16 per layer, Lk averaged over the steps), + the logits written and read back by the argmax; against 8 TB/s (6.29 achievable)
        H, NL, V = cfg.hidden_size, cfg.num_hidden_layers, cfg.vocab_size
        w_bytes = 2 * (sum(p.numel() for n, p in m.named_parameters() if n.startswith("bert.encoder.")) + H * H + V * H)
        lk_avg = sum(in_len + s + 1 for s in range(1, T)) / max(T - 1, 1)
        kv_bytes = B * lk_avg * 2 * H * 2 * NL
        logit_bytes = 2 * B * V * 2
        step_bytes = w_bytes + kv_bytes + logit_bytes
        # the first step (whole prefix, 102 rows per sequence) is a different regime: time it alone and subtract
        first_args = (img, vis_pe, input_ids[:, :in_len], token_type[:, :in_len + 1].contiguous(), pos[:, :in_len + 1].contiguous(),
                      am[:, :in_len + 1, :in_len + 1].contiguous())
        for _ in range(2):
            m(*first_args)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            m(*first_args)
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t1) / n * 1e3
        tok_ms = max(dt * 1e3 - first_ms, 0.0) / max(T - 1, 1)
        out[name].update({"first_step_ms": round(first_ms, 3), "ms_per_token_step_excl_first": round(tok_ms, 4)})
        out["roofline"] = {"bound": "hbm", "kernel": "one greedy token step (all launches of the step; B = %d, history %.0f positions)" % (B, lk_avg),
                           "achieved": round(step_bytes / (tok_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                           "frac": round(step_bytes / (tok_ms * 1e-3) / 8e12, 4), "traffic": None,
                           "algorithmic_mb": {"weights": round(w_bytes / 1e6, 1), "kv_history": round(kv_bytes / 1e6, 1), "logits": round(logit_bytes / 1e6, 1)},
                           "floor_ms_at_6.29_TBps": round(step_bytes / 6.29e12 * 1e3, 4)}
    del m
    torch.cuda.empty_cache()          # (workspaces + captured graphs of one mode otherwise fragment the next mode's allocations: beam 5 after greedy + beam 3 measured 90 ms instead of 34)
print(json.dumps(out))
if os.environ.get("VLP_DEBUG_TUNE") == "1":
    from vlp_amd.engine import Engine
    print("skinny choices:", sorted(Engine._skinny_choice.items()), file=sys.stderr)
