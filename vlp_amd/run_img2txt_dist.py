"""MI355X drop-in for `vlp/run_img2txt_dist.py` (the reference's fine-tuning / pre-training entry point).

Same command line as the reference (every flag of run_img2txt_dist.py:47-188 is accepted with the same
default), same per-process launch style (`--local_rank k --global_rank k --world_size N`, file:// rendezvous,
README.md:139-155) and same step loop (:453-586): forward -> summed loss -> (scaled) backward -> warmup_linear
learning rate -> optimizer step -> zero_grad, per-epoch checkpoint `model.{epoch}.bin` on rank 0 (:588-599),
barrier (:604-605).  What differs is underneath: the model / optimizer / DDP come from vlp_amd (HIP kernels
behind libvlp_hip.so, RCCL gradient buckets) instead of torch autograd + apex + torch DDP.

Data: the reference's loader (vlp/seq2seq_loader.py) needs h5py/torchvision and the COCO/CC/VQA feature
files, none of which exist on the build or bench machines; it is the ranked-next row N3 of SURVEY.md 8(f).
`--synthetic STEPS_PER_EPOCH` feeds seeded synthetic batches that follow the same 12-tuple contract
(vlp_amd/synthetic.py); without it the script stops with a clear message.

Launched by torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment) it takes ranks from there.
"""
import argparse
import copy
import glob
import json
import logging
import math
import os
import random
import time
from pathlib import Path

import numpy as np
import torch

from . import synthetic
from .distributed import DistributedDataParallel as DDP
from .modeling import BertConfig, BertForPreTrainingLossMask, load_checkpoint_state
from .optimization import BertAdam, warmup_linear           # noqa: F401  (re-exported like the reference imports them)
from .optimization_fp16 import FP16_Optimizer_State, FusedAdam

KNOWN_VOCABS = {"bert-base-cased": 28996, "bert-large-cased": 28996, "bert-base-uncased": 30522, "bert-large-uncased": 30522}


def _get_max_epoch_model(output_dir):   # reference :33-43
    fn_model_list = glob.glob(os.path.join(output_dir, "model.*.bin"))
    fn_optim_list = glob.glob(os.path.join(output_dir, "optim.*.bin"))
    if (not fn_model_list) or (not fn_optim_list):
        return None
    both = set(int(Path(fn).stem.split(".")[-1]) for fn in fn_model_list) & set(int(Path(fn).stem.split(".")[-1]) for fn in fn_optim_list)
    return max(both) if both else None


def _to_cpu(x):
    if torch.is_tensor(x):
        return x.detach().cpu()
    if isinstance(x, dict):
        return {k: _to_cpu(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_to_cpu(v) for v in x)
    return x


def build_parser():
    p = argparse.ArgumentParser()
    # General (same names / defaults as the reference)
    p.add_argument("--bert_model", default="bert-base-cased", type=str)
    p.add_argument("--config_path", default=None, type=str)
    p.add_argument("--output_dir", default="tmp", type=str)
    p.add_argument("--log_file", default="training.log", type=str)
    p.add_argument("--model_recover_path", default=None, type=str)
    p.add_argument("--do_train", action="store_true")
    p.add_argument("--do_lower_case", action="store_true")
    p.add_argument("--train_batch_size", default=64, type=int)
    p.add_argument("--learning_rate", default=3e-5, type=float)
    p.add_argument("--label_smoothing", default=0, type=float)
    p.add_argument("--weight_decay", default=0.01, type=float)
    p.add_argument("--finetune_decay", action="store_true")
    p.add_argument("--num_train_epochs", default=30, type=int)
    p.add_argument("--warmup_proportion", default=0.1, type=float)
    p.add_argument("--no_cuda", action="store_true")
    p.add_argument("--local_rank", type=int, default=-1)
    p.add_argument("--global_rank", type=int, default=-1)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--fp16", action="store_true")
    p.add_argument("--fp32_embedding", action="store_true")
    p.add_argument("--loss_scale", type=float, default=0)
    p.add_argument("--amp", action="store_true")
    p.add_argument("--from_scratch", action="store_true")
    p.add_argument("--new_segment_ids", action="store_true")
    p.add_argument("--tokenized_input", action="store_true")
    p.add_argument("--len_vis_input", type=int, default=100)
    p.add_argument("--max_len_b", type=int, default=20)
    p.add_argument("--trunc_seg", default="b")
    p.add_argument("--always_truncate_tail", action="store_true")
    p.add_argument("--mask_prob", default=0.15, type=float)
    p.add_argument("--max_pred", type=int, default=3)
    p.add_argument("--num_workers", default=4, type=int)
    p.add_argument("--max_position_embeddings", type=int, default=None)
    # Others for VLP
    p.add_argument("--src_file", default=["/mnt/dat/COCO/annotations/dataset_coco.json"], type=str, nargs="+")
    p.add_argument("--enable_visdom", action="store_true")
    p.add_argument("--visdom_port", type=int, default=8888)
    p.add_argument("--image_root", type=str, default="/mnt/dat/COCO/images")
    p.add_argument("--dataset", default="coco", type=str)
    p.add_argument("--split", type=str, nargs="+", default=["train", "restval"])
    p.add_argument("--world_size", default=1, type=int)
    p.add_argument("--dist_url", default="file://[PT_OUTPUT_DIR]/nonexistent_file", type=str)
    p.add_argument("--file_valid_jpgs", default="/mnt/dat/COCO/annotations/coco_valid_jpgs.json", type=str)
    p.add_argument("--sche_mode", default="warmup_linear", type=str)
    p.add_argument("--drop_prob", default=0.1, type=float)
    p.add_argument("--use_num_imgs", default=-1, type=int)
    p.add_argument("--vis_mask_prob", default=0, type=float)
    p.add_argument("--max_drop_worst_ratio", default=0, type=float)
    p.add_argument("--drop_after", default=6, type=int)
    p.add_argument("--s2s_prob", default=1, type=float)
    p.add_argument("--bi_prob", default=0, type=float)
    p.add_argument("--enable_butd", action="store_true")
    p.add_argument("--region_bbox_file", default="coco_detection_vg_thresh0.2_feat_gvd_checkpoint_trainvaltest.h5", type=str)
    p.add_argument("--region_det_file_prefix", default="feat_cls_1000/coco_detection_vg_100dets_gvd_checkpoint_trainval", type=str)
    p.add_argument("--tasks", default="img2txt")
    p.add_argument("--relax_projection", action="store_true")
    p.add_argument("--scst", action="store_true")
    # vlp_amd additions
    p.add_argument("--synthetic", type=int, default=0, metavar="STEPS_PER_EPOCH",
                   help="train on seeded synthetic batches (vlp_amd/synthetic.py) for this many steps per epoch")
    p.add_argument("--packed_features", default="", help="directory of a vlp_amd.data packed region store (write_packed / pack_from_h5)")
    p.add_argument("--token_file", default="", help="json list of [image id, [caption token ids]] (captions pre-tokenised with the "
                                                     "reference's WordPiece vocabulary); used with --packed_features")
    p.add_argument("--num_hidden_layers", type=int, default=None, help="override the config's depth (plumbing tests)")
    p.add_argument("--optim_format", default="vlp", choices=["vlp", "apex"],
                   help="layout of optim.N.bin: 'vlp' = the engine's flat buffers (+ dropout stream position: bit-exact resume), 'apex' = the "
                        "reference stack's FP16_Optimizer(FusedAdam) state_dict, loadable by the reference; both are accepted on resume")
    p.add_argument("--stop_after_epoch", type=int, default=0, help="stop after this epoch (schedule still spans --num_train_epochs); 0 = off")
    p.add_argument("--log_every", type=int, default=100, help="steps between loss read-backs (each read-back is a host sync)")
    p.add_argument("--allow_fp16_compute", action="store_true",
                   help="without --fp16 the reference trains in fp32 under BertAdam (run_img2txt_dist.py:421-426; its README's `--amp` commands do "
                        "exactly that, because amp only engages together with --fp16, :305).  vlp_amd has no fp32 compute path: this switch runs "
                        "that command line with fp16 storage / fp32 accumulate and BertAdam on fp32 master weights (static loss scale --loss_scale, "
                        "default 1024)")
    p.add_argument("--no_padding_free", action="store_true",
                   help="compute all L positions of every sample like the reference does; by default positions past a sample's last token "
                        "(attended by nothing, read by no loss) are not computed when the batch says where they start (DESIGN.md section 7)")
    p.add_argument("--shard_order", default="balanced", choices=["balanced", "reference"],
                   help="world_size > 1 with --packed_features: 'balanced' deals every global batch (DistributedSampler's sample set) to the ranks "
                        "by caption length so that padding-free row counts per rank match; 'reference' is DistributedSampler's index order bit for bit")
    return p


def derive_args(args):
    """Derived fields + the reference's argument checks (:193-209, :238-243)."""
    if "RANK" in os.environ and args.global_rank == -1:     # torchrun style launch
        args.global_rank = int(os.environ["RANK"])
        args.local_rank = int(os.environ.get("LOCAL_RANK", 0))
        args.world_size = int(os.environ.get("WORLD_SIZE", 1))
        if args.dist_url.startswith("file://[PT_OUTPUT_DIR]"):
            args.dist_url = "env://"
    args.max_seq_length = args.max_len_b + args.len_vis_input + 3     # +3 for 2x[SEP] and [CLS]
    args.mask_image_regions = args.vis_mask_prob > 0
    args.dist_url = args.dist_url.replace("[PT_OUTPUT_DIR]", args.output_dir)
    assert args.tasks in ("img2txt", "vqa2")
    assert args.enable_butd is True, "only support region attn! featmap attn deprecated"
    if args.scst:
        raise NotImplementedError("--scst needs the coco-caption CIDEr scorer (empty submodule in the reference checkout); out of scope")
    if args.gradient_accumulation_steps < 1:
        raise ValueError("Invalid gradient_accumulation_steps parameter: {}, should be >= 1".format(args.gradient_accumulation_steps))
    args.train_batch_size = int(args.train_batch_size / args.gradient_accumulation_steps)
    return args


def model_config(args):
    cfg_file = args.config_path or (os.path.join(args.bert_model, "bert_config.json") if os.path.isdir(args.bert_model) else None)
    if cfg_file:
        config = BertConfig.from_json_file(cfg_file)
    elif args.bert_model in KNOWN_VOCABS:
        large = "large" in args.bert_model
        config = BertConfig(KNOWN_VOCABS[args.bert_model], hidden_size=1024 if large else 768, num_hidden_layers=24 if large else 12,
                            num_attention_heads=16 if large else 12, intermediate_size=4096 if large else 3072)
    else:
        raise EnvironmentError("--bert_model must be a local directory with bert_config.json (no network)")
    config.type_vocab_size = 6 if args.new_segment_ids else 2
    config.hidden_dropout_prob = config.attention_probs_dropout_prob = args.drop_prob
    if args.max_position_embeddings:
        config.max_position_embeddings = args.max_position_embeddings
    if args.num_hidden_layers:
        config.num_hidden_layers = args.num_hidden_layers
    for key, default in (("relax_projection", 0), ("task_idx", None), ("fp32_embedding", False), ("label_smoothing", None)):
        if not hasattr(config, key):
            setattr(config, key, default)
    return config


def build_model(args, device):
    """run_img2txt_dist.py:310-377: construct (from scratch or from a checkpoint), .half(), .to(device)."""
    if args.relax_projection:
        raise NotImplementedError("--relax_projection is not supported by vlp_amd")
    if not args.fp16 and not args.allow_fp16_compute:
        raise NotImplementedError(
            "vlp_amd implements the reference's --fp16 path (fp16 storage, fp32 accumulate, FusedAdam in FP16_Optimizer_State) and has no fp32 compute "
            "path.  This command line has no --fp16%s.  Add --fp16 (the reference's fp16 recipe), or --allow_fp16_compute to keep the optimizer this "
            "command line selects (BertAdam, run_img2txt_dist.py:421-426) on fp32 master weights with fp16 compute."
            % (": the reference README's example commands pass --amp alone (README.md:112,131), and amp only engages together with --fp16 "
               "(run_img2txt_dist.py:305), so the reference itself runs them in fp32" if args.amp else ""))
    config = model_config(args)
    state = None
    if args.model_recover_path:
        state = torch.load(args.model_recover_path, map_location="cpu")
    elif not args.from_scratch:
        wpath = os.path.join(args.bert_model, "pytorch_model.bin") if os.path.isdir(args.bert_model) else None
        if not wpath or not os.path.exists(wpath):
            raise EnvironmentError("no pretrained weights available offline: pass --from_scratch or --model_recover_path")
        state = torch.load(wpath, map_location="cpu")
    model = BertForPreTrainingLossMask(config, num_labels=2, enable_butd=args.enable_butd, len_vis_input=args.len_vis_input,
                                       tasks=args.tasks, allow_random_fc7=bool(args.synthetic))
    if state is not None:
        # the reference always goes through from_pretrained(state_dict=...) (:324-336): gamma/beta renames, segment table 2 -> 6
        # rows for --new_segment_ids, position-table tiling for --max_position_embeddings, and an ERROR on any other size mismatch
        load_checkpoint_state(model, state)
    model.half()
    model.to(device)
    return model


def build_optimizer(args, model, t_total):
    """run_img2txt_dist.py:393-426: two param groups by name, FusedAdam inside the fp16 wrapper."""
    named = list(model.named_parameters())
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    if not args.fp16:        # --allow_fp16_compute: the reference's else-branch (:421-426) on fp32 masters of the fp16 parameters
        from .optimization import BertAdam
        opt = BertAdam(groups, lr=args.learning_rate, warmup=args.warmup_proportion, schedule=args.sche_mode, t_total=t_total)
        opt.grad_scale = float(args.loss_scale) if args.loss_scale > 0 else 1024.0     # fp16 backward needs a scaled loss; step() divides it out
        return opt
    inner = FusedAdam(groups, lr=args.learning_rate, bias_correction=False, max_grad_norm=1.0)
    if args.loss_scale == 0:
        return FP16_Optimizer_State(inner, dynamic_loss_scale=True)
    return FP16_Optimizer_State(inner, static_loss_scale=args.loss_scale)


def train_step(model, optimizer, batch, lr_this_step, mask_image_regions=False, drop_worst_ratio=0.0, accumulate=False, accum_steps=1):
    """One iteration of the reference's inner loop (:479-585) on device-resident tensors.  Returns the (un-normalised) loss
    tuple (device tensors; nothing is read back).  With gradient accumulation the back-propagated loss is divided by
    `accum_steps` exactly as the reference does before `optimizer.backward` (:567-571)."""
    (input_ids, segment_ids, input_mask, lm_label_ids, masked_pos, masked_weights, is_next, task_idx, img, vis_masked_pos, vis_pe,
     ans_labels) = batch
    loss_tuple = model(img, vis_pe, input_ids, segment_ids, input_mask, lm_label_ids, ans_labels, is_next, masked_pos=masked_pos,
                       masked_weights=masked_weights, task_idx=task_idx, vis_masked_pos=vis_masked_pos,
                       mask_image_regions=mask_image_regions, drop_worst_ratio=drop_worst_ratio)
    masked_lm_loss, pretext_loss, ans_loss = loss_tuple
    # :531 `loss = masked_lm_loss + pretext_loss + ans_loss`; the engine's shared zero placeholder (a loss this task does not have) is
    # recognised by identity and not added: no add launches for `+ 0`
    eng = getattr(model.module if hasattr(model, "module") else model, "engine", None)
    terms = [t for t in loss_tuple if eng is None or not eng.is_zero_placeholder(t)]
    if not terms:
        raise RuntimeError("train_step: this batch has no live loss (no masked positions, no answer labels, no masked regions): nothing to back-propagate")
    loss = terms[0]
    for t in terms[1:]:
        loss = loss + t
    if accum_steps > 1:
        loss = loss / accum_steps                            # :567-568
    if hasattr(optimizer, "backward"):
        optimizer.backward(loss)                             # :571
    else:                                                    # :573 `loss.backward()` (BertAdam; --allow_fp16_compute scales the fp16 backward)
        gs = getattr(optimizer, "grad_scale", 1.0)
        loss.float().backward(gradient=torch.full_like(loss, gs, dtype=torch.float32) if gs != 1.0 else None)
    if not accumulate:
        if hasattr(optimizer, "backward"):
            for g in optimizer.param_groups:                 # :580-583 (fp16 only: BertAdam runs its own schedule)
                g["lr"] = lr_this_step
        optimizer.step()
        optimizer.zero_grad()
    return loss_tuple


def build_packed_loader(args, device):
    """Img2txtDataset + Preprocess4Seq2seq (vlp/seq2seq_loader.py:62-359, run_img2txt_dist.py:248-300) on a packed region store:
    s2s / bidirectional preprocessors drawn per sample with --s2s_prob / --bi_prob.  With world_size > 1 the per-epoch order is
    DistributedSampler's (:295, 455): one global permutation per epoch (seeded by the epoch, the same on every rank), padded by
    wrapping around so that every rank gets ceil(N / W) samples, rank r taking every W-th element (vlp_amd.data.distributed_sampler_indices)."""
    from .data import BatchPrefetcher, PackedRegionStore, TextPreprocessor
    store = PackedRegionStore(args.packed_features)
    with open(args.token_file) as f:
        examples = [(e[0], e[1]) for e in json.load(f)]
    kw = dict(max_pred=args.max_pred, mask_prob=args.mask_prob, vocab_size=KNOWN_VOCABS.get(args.bert_model, 28996), cls_id=synthetic.CLS_ID,
              sep_id=synthetic.SEP_ID, mask_id=synthetic.MASK_ID, unk_id=synthetic.UNK_ID, max_len=args.max_seq_length, max_len_b=args.max_len_b,
              len_vis_input=args.len_vis_input, new_segment_ids=args.new_segment_ids, trunc_seg=args.trunc_seg,
              always_truncate_tail=args.always_truncate_tail)
    return BatchPrefetcher(store, examples, args.train_batch_size, TextPreprocessor(mode="s2s", **kw), TextPreprocessor(mode="bi", **kw),
                           s2s_prob=args.s2s_prob, device=device, seed=args.seed, vis_mask_prob=args.vis_mask_prob,
                           rank=max(args.global_rank, 0), world=max(args.world_size, 1), num_workers=args.num_workers,
                           balance_lengths=(args.shard_order == "balanced" and not args.no_padding_free))


def synthetic_batches(args, device, steps, rank):
    """Device-resident synthetic batches (a small rotating pool, seeded per rank like a DistributedSampler shard).  The pool is built once per
    run and reused by every epoch: the same tensor objects come back, so the padding-free step derives their kept lengths once."""
    pool = getattr(args, "_synthetic_pool", None)
    if pool is not None:
        for s in range(steps):
            yield pool[s % len(pool)]
        return
    pool = args._synthetic_pool = []
    for i in range(min(4, steps)):
        b = synthetic.make_batch(args.train_batch_size, max_len_b=args.max_len_b, len_vis_input=args.len_vis_input,
                                 vocab_size=KNOWN_VOCABS.get(args.bert_model, 28996), max_pred=args.max_pred, mask_prob=args.mask_prob,
                                 s2s_prob=args.s2s_prob, tasks=args.tasks, seed=args.seed + 1000 * max(rank, 0) + i,
                                 new_segment_ids=args.new_segment_ids, vis_mask_prob=args.vis_mask_prob)
        pool.append(synthetic.batch_to(b, device, half=True))
    for s in range(steps):
        yield pool[s % len(pool)]


def main(argv=None):
    args = derive_args(build_parser().parse_args(argv))
    os.makedirs(args.output_dir, exist_ok=True)
    json.dump(args.__dict__, open(os.path.join(args.output_dir, "opt.json"), "w"), sort_keys=True, indent=2)
    logging.basicConfig(filename=os.path.join(args.output_dir, args.log_file), filemode="w",
                        format="%(asctime)s - %(levelname)s - %(name)s -   %(message)s", datefmt="%m/%d/%Y %H:%M:%S", level=logging.INFO,
                        force=True)      # (a second main() in one process logs to ITS output_dir, not to the first one's file)
    logger = logging.getLogger(__name__)
    if args.no_cuda or not torch.cuda.is_available():
        raise RuntimeError("vlp_amd has no CPU path: an MI355X is required (the reference's --no_cuda mode is not provided)")
    distributed = args.local_rank != -1
    if distributed:
        torch.cuda.set_device(args.local_rank)
        device = torch.device("cuda", args.local_rank)
        torch.distributed.init_process_group(backend="nccl", init_method=args.dist_url, world_size=args.world_size, rank=args.global_rank)
    else:
        device = torch.device("cuda")
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed_all(args.seed)

    loader = None
    if args.packed_features:
        loader = build_packed_loader(args, device)
        steps_per_epoch = loader.steps
    elif args.synthetic:
        steps_per_epoch = args.synthetic
    else:
        raise NotImplementedError("give --packed_features DIR --token_file FILE (vlp_amd.data; the reference's h5 files are converted once "
                                  "with vlp_amd.data.pack_from_h5) or --synthetic STEPS_PER_EPOCH; the tokenizer itself is out of scope")
    t_total = int(steps_per_epoch * args.num_train_epochs * 1. / args.gradient_accumulation_steps)

    recover_step = _get_max_epoch_model(args.output_dir)     # :310: resume from the newest epoch that has model AND optimizer files
    if recover_step:
        logger.info("***** Recover model: %d *****", recover_step)
        args.model_recover_path = os.path.join(args.output_dir, "model.{0}.bin".format(recover_step))
    model = build_model(args, device)
    if distributed:
        model = DDP(model, device_ids=[args.local_rank], output_device=args.local_rank, find_unused_parameters=True)
    optimizer = build_optimizer(args, model, t_total)
    eng = (model.module if hasattr(model, "module") else model).engine
    if args.no_padding_free:
        eng.varlen = False
    elif eng.varlen == "auto":
        # padding-free by default: the packed loader's MaskSpec carries the lengths on the host; a --synthetic pool of dense
        # masks is reduced + read back once per pooled tensor (budget), then remembered; anything else runs dense and says so once
        eng.varlen_readback_budget = 8
    logger.info("padding-free step: %s", eng.varlen)
    if hasattr(optimizer, "pipeline_with_forward"):
        # this loop touches parameters only through the engine, so the optimizer step MAY stream underneath the next forward
        # (VLP_ADAM_PIPELINE=1; bit-identical).  On one MI355X it is a wash (the HBM-bound update slows the concurrent GEMMs), so off.
        optimizer.pipeline_with_forward = os.environ.get("VLP_ADAM_PIPELINE", "0") == "1"
    global_step = 0
    if recover_step:                                         # :428-437
        logger.info("***** Recover optimizer: %d *****", recover_step)
        optimizer.load_state_dict(torch.load(os.path.join(args.output_dir, "optim.{0}.bin".format(recover_step)), map_location=device))
        if args.loss_scale == 0:
            optimizer.dynamic_loss_scale = True
        global_step = math.floor(recover_step * t_total * 1. / args.num_train_epochs)      # :337-338
    logger.info("***** Running training *****  batch %d, steps %d", args.train_batch_size, t_total)
    model.train()
    stop_after = args.stop_after_epoch if args.stop_after_epoch > 0 else args.num_train_epochs
    for i_epoch in range((recover_step or 0) + 1, min(args.num_train_epochs, stop_after) + 1):
        t0 = time.time()
        losses = []
        if loader is not None:
            loader.set_epoch(i_epoch - 1)                                                             # train_sampler.set_epoch(i_epoch-1), :455
        for step, batch in enumerate(loader if loader is not None else synthetic_batches(args, device, steps_per_epoch, args.global_rank)):
            acc = (step + 1) % args.gradient_accumulation_steps != 0
            lr = args.learning_rate * warmup_linear(global_step / t_total, args.warmup_proportion)
            lt = train_step(model, optimizer, batch, lr, mask_image_regions=args.mask_image_regions,
                            drop_worst_ratio=args.max_drop_worst_ratio if i_epoch > args.drop_after else 0,
                            accumulate=acc, accum_steps=args.gradient_accumulation_steps)
            if not acc:
                global_step += 1
            if step % args.log_every == 0:        # the only host read-back; the reference does 4 per step (:535-538)
                losses.append(float((lt[0] + lt[1] + lt[2]).detach()))
                logger.info("Epoch %d, Iter %d, Loss %.3f", i_epoch, step, losses[-1])
        torch.cuda.synchronize()
        dt = time.time() - t0
        if hasattr(optimizer, "consolidate"):
            optimizer.consolidate()        # collective under VLP_DDP_MODE=sharded (every rank; rank 0 alone writes the file below)
        if args.global_rank in (-1, 0):
            print("epoch %d: %d steps, %.1f samples/s/rank, loss %s" % (i_epoch, steps_per_epoch, steps_per_epoch * args.train_batch_size / dt,
                                                                        ["%.3f" % l for l in losses[-3:]]))
            to_save = model.module if hasattr(model, "module") else model
            torch.save(copy.deepcopy(to_save).cpu().state_dict(), os.path.join(args.output_dir, "model.{0}.bin".format(i_epoch)))
            # the reference disabled this line ("need to sanitize state and ship everything back to cpu", :599); here the optimizer
            # state is three flat fp32 buffers per group + a few scalars, shipped to the host as they are
            osd = optimizer.apex_state_dict() if (args.optim_format == "apex" and hasattr(optimizer, "apex_state_dict")) else optimizer.state_dict()
            torch.save(_to_cpu(osd), os.path.join(args.output_dir, "optim.{0}.bin".format(i_epoch)))
        if args.world_size > 1:
            torch.distributed.barrier()
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
