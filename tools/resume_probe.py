"""Where does a resumed run diverge?  3 steps, snapshot, 1 more step  vs  fresh objects + snapshot + the same step."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vlp_amd import run_img2txt_dist as R  # noqa: E402
from vlp_amd import synthetic as S  # noqa: E402

dev = torch.device("cuda")
args = R.derive_args(R.build_parser().parse_args(["--bert_model", "bert-base-cased", "--from_scratch", "--fp16", "--enable_butd", "--len_vis_input", "100",
                                                  "--new_segment_ids", "--synthetic", "3", "--train_batch_size", "4", "--max_len_b", "20",
                                                  "--num_hidden_layers", "2", "--loss_scale", "0", "--output_dir", "/tmp/x"]))
torch.manual_seed(7)
model = R.build_model(args, dev).train()
opt = R.build_optimizer(args, model, 6)
batches = list(R.synthetic_batches(args, dev, 4, 0))
for i in range(3):
    R.train_step(model, opt, batches[i], 3e-4)
msd = copy.deepcopy(model).cpu().state_dict()
osd = R._to_cpu(opt.state_dict())
eng = model.engine
snap = {k: v.clone() for k, v in eng.flat.items()}
snap_p = {n: eng.P(n).clone() for n in eng._params}
snap_m = [t.clone() for t in opt.fp32_groups_flat]
scale = opt._scale_state.clone()
R.train_step(model, opt, batches[3], 3e-4)
ref = {k: v.clone() for k, v in eng.flat.items()}

torch.manual_seed(99)
args.model_recover_path = None
m2 = R.build_model(args, dev)
m2.load_state_dict(msd, strict=False)
m2 = m2.half().to(dev).train()
o2 = R.build_optimizer(args, m2, 6)
o2.load_state_dict({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in osd.items()} if False else torch.load(__import__("io").BytesIO(
    (lambda b: (torch.save(osd, b), b.getvalue())[1])(__import__("io").BytesIO())), map_location=dev))
e2 = m2.engine
for k in snap:
    print("flat", k, "equal before step:", torch.equal(snap[k], e2.flat[k]))
bad = [(n, float((snap_p[n].float() - e2.P(n).float()).abs().max())) for n in snap_p if not torch.equal(snap_p[n], e2.P(n))]
print("params differing before step:", len(bad), bad[:8])
for a, b in zip(snap_m, o2.fp32_groups_flat):
    print("master equal:", torch.equal(a, b))
print("scale state", scale.tolist(), o2._scale_state.tolist(), "rng", eng.step_seed - 1, e2.step_seed)
R.train_step(m2, o2, batches[3], 3e-4)
for k in ref:
    print("flat", k, "equal after step:", torch.equal(ref[k], e2.flat[k]), float((ref[k].float() - e2.flat[k].float()).abs().max()))
