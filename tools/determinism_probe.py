"""Runs the synthetic entry script twice from scratch (and once split by a resume) and reports which tensors differ."""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vlp_amd import run_img2txt_dist as R  # noqa: E402

common = ["--bert_model", "bert-base-cased", "--from_scratch", "--fp16", "--enable_butd", "--len_vis_input", "100", "--new_segment_ids",
          "--synthetic", "3", "--num_train_epochs", "2", "--train_batch_size", "4", "--max_len_b", "20", "--num_hidden_layers", "2",
          "--learning_rate", "3e-4", "--warmup_proportion", "0.3", "--loss_scale", "0", "--seed", "7"] + sys.argv[1:]
with tempfile.TemporaryDirectory() as d:
    a, a2, b = (os.path.join(d, x) for x in ("a", "a2", "b"))
    R.main(common + ["--output_dir", a])
    R.main(common + ["--output_dir", a2])
    R.main(common + ["--output_dir", b, "--stop_after_epoch", "1"])
    R.main(common + ["--output_dir", b])
    for tag, x, y in (("scratch vs scratch, epoch 1", a + "/model.1.bin", a2 + "/model.1.bin"), ("scratch vs scratch, epoch 2", a + "/model.2.bin", a2 + "/model.2.bin"),
                      ("scratch vs split, epoch 1", a + "/model.1.bin", b + "/model.1.bin"), ("scratch vs resumed, epoch 2", a + "/model.2.bin", b + "/model.2.bin")):
        sx, sy = torch.load(x), torch.load(y)
        bad = [(k, float((sx[k].float() - sy[k].float()).abs().max())) for k in sx if not torch.equal(sx[k], sy[k])]
        print(tag, ": %d of %d tensors differ" % (len(bad), len(sx)), bad[:6])
