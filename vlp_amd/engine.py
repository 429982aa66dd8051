"""Fused forward/backward executor of the VLP unified transformer on MI355X.

This is the host-side orchestration that replaces the reference's per-op autograd graph
(pytorch_pretrained_bert/modeling.py:217-482, 1033-1143): one Python call sequence over the C ABI of
libvlp_hip.so per training step, with

  * parameters living in two flat fp16 buffers (weight-decay group / no-decay group -- the same two
    groups the train script builds, run_img2txt_dist.py:394-401), laid out in *backward completion
    order* so that a gradient bucket is a contiguous slice that can be handed to RCCL the moment the
    last weight-gradient GEMM of that slice has been launched;
  * gradients written by the kernels straight into matching flat fp16 buffers (`param.grad` are views);
  * activations kept in a per-shape workspace that is reused every step (3.4 GB at B = 64, L = 167 -- small against the
    chip's 288 GB, so nothing is recomputed except dropout masks, which are a pure function of (seed, stream, index));
  * transposed weight shadows for the dgrad GEMMs refreshed once per backward.

PyTorch is used for memory (torch.empty), streams and the autograd hand-off only.  There is no
fallback: every compute step is a HIP kernel.
"""
import math
import os
import weakref

import numpy as np
import torch

from . import _lib as K
from . import tuning
from .input_prep import MaskSpec, RawRegions

ALIGN = 64            # elements; every parameter starts on a 128-byte boundary inside its flat buffer
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")    # run_img2txt_dist.py:395
PE_DIM = 1607         # 6 box numbers + 1601 class probabilities (modeling.py:1016)
PE_PAD = 1664         # next multiple of 64: GEMM K alignment


def _ru(x, m):
    return (x + m - 1) // m * m


def is_no_decay(name):
    return any(nd in name for nd in NO_DECAY)


class VlpPerformanceWarning(UserWarning):
    """A correct but avoidably slow way of driving the engine (issued once per engine)."""


class _State(object):
    """What one forward leaves behind for its backward."""
    __slots__ = ("gen", "B", "L", "P", "seed", "p_drop", "ws", "batch", "task", "has_mlm", "task_labels", "pretext", "pk")


class Engine(object):
    GEMM_NT_VARIANT = None   # None -> vlp_amd.tuning (committed table, else shape heuristic; timing search only with VLP_AUTOTUNE=1); or force an int
    NT_CANDIDATES = (1, 2, 4, 5, 9, 10, 11, 12, 13, 27, 29, 73, 77, 264)   # product variants (+8 = XCD-aware tile order, +16 = ring, 64 + cfg = wave-pipelined, 256 = persistent k stream); see include/vlp_hip.h
    NT_CANDIDATES_SKINNY = (1, 2, 9, 10, 11, 17)      # M <= 1024 (decoding, LM head): few workgroups, latency-bound -> also the 4-stage ring
    GEMM_TN_VARIANT = 2      # ds_read_b64_tr_b16 fragment reads + LDS-DMA staging
    # weight-gradient GEMMs of the encoder layers on a second HIP stream, concurrent with the dgrad chain (the wgrad grids are a single
    # round of 430-580 workgroups whose tails leave CUs idle; so do LN / attention backward): +2.6 % step throughput on one MI355X.
    # Results are bit-identical with and without (tests/test_10_model_gpu.py).  VLP_WGRAD_SIDE_STREAM=0 turns it off (clean
    # single-kernel timings under rocprofv3).  bench.py's live roofline samples stay single-kernel measurements: a sampled launch
    # first lets the side stream drain (see _nt).
    WGRAD_SIDE_STREAM = os.environ.get("VLP_WGRAD_SIDE_STREAM", "1") == "1"
    LN_DEFER = os.environ.get("VLP_LN_DEFER", "1") == "1"               # LayerNorm dgamma / dbeta second stages batched into one launch per backward
    TAIL_ON_SIDE = os.environ.get("VLP_TAIL_SIDE", "1") == "1"          # embedding-table gradients + that batched launch on the side stream, under the region-projection backward
    SHADOW_ON_SIDE = os.environ.get("VLP_SHADOW_SIDE", "1") == "1"      # W^T shadows transposed on the side stream during the forward
    # mask_image_regions: the reference's loader line `input_mask[:, vis_masked_pos].fill_(0)` (seq2seq_loader.py:303-304) indexes with a
    # numpy array -- advanced indexing, i.e. it fills a COPY and leaves the mask untouched (checked on torch 2.10 and, against the unmodified
    # loader, by the reference-pinning CPU tests): masked regions enter the encoder as zeros but stay attendable.
    # Parity follows that behaviour; VLP_BLOCK_MASKED_REGIONS=1 follows the line's comment instead ("block the masked visual feature")
    # on the MaskSpec path (a dense attention_mask is always used as given).
    BLOCK_MASKED_REGION_KEYS = os.environ.get("VLP_BLOCK_MASKED_REGIONS", "0") == "1"
    # Padding-free (packed) step, opt-in (VLP_VARLEN=1 or engine.varlen = True; training / grad-enabled forwards only).  A sample's
    # positions past its last token are INERT in the reference's computation: no kept query attends them (their mask columns are 0 for
    # every row: seq2seq_loader.py:295-304), no loss reads them (masked_pos, h[:, 0], h[:, Nv+1] lie before them), so their hidden states
    # feed nothing and every gradient contribution through them is exactly zero (dO = 0 for pad queries, P = 0 for pad keys:
    # modeling.py:289-298).  The packed step keeps the first n_b positions of sample b (n_b derived from the mask itself, _packing())
    # as rows [row_off[b], row_off[b+1]) of every [M, *] activation and runs every row-wise kernel on M' = sum n_b rows; attention takes
    # row_off; dropout hashes keep the logical (b*L + l, col) element, so masks -- and with them losses, logits and gradients --
    # equal the dense run's up to the fp32 summation order of the weight-gradient / LayerNorm-parameter sums.
    # Round 6: ON BY DEFAULT where it costs nothing -- VLP_VARLEN unset = "auto": a batch whose kept lengths are known on the host (MaskSpec.lens_host:
    # the loader's form) or remembered for this very mask tensor (a device-resident pool) runs packed; a FRESH dense int64 mask would need a
    # device -> host read-back before the forward can be enqueued (a pipeline bubble per step), so in auto mode it runs dense after
    # `varlen_readback_budget` read-backs (0 here; the entry script allows 8 so that a --synthetic pool warms up) and a one-time
    # VlpPerformanceWarning says so.  VLP_VARLEN=1 / engine.varlen = True: always packed, read-backs accepted (warned about once);
    # VLP_VARLEN=0 / False: never.
    VARLEN = {"1": True, "0": False}.get(os.environ.get("VLP_VARLEN", ""), "auto")
    # (Round 5, measured and NOT kept -- the gradient norm in front of the optimizer step stays one vlp_sumsq pass over the buffer, 53 us:
    # one partial-sum launch per finished gradient slice on the side stream during backward LOSES, 9.738 / 9.722 vs 9.666 / 9.671 ms/step;
    # summing the 93 % of the buffer that is final when the side stream reaches the embedding tables, under the region-projection backward,
    # is a wash, 9.568 / 9.548 vs 9.558 / 9.573 -- profiles/r05_instep_ab_norm_per_slice.txt.  Both forms were removed.)
    GROUPED_WGRAD = os.environ.get("VLP_GROUPED_WGRAD", "1") == "1"     # one vlp_gemm_tn_grouped launch per layer instead of 4 split-M wgrads + 4 reduces
    TN_SPLITS = None         # None -> vlp_amd.tuning (variant flags, split-M factor) per (M, N, K)
    # split-M factor: the wgrad outputs are small (36..144 tiles of 128x128) and the contraction long (M = 10 688), so the
    # workgroup count tiles*splits has to land just under a multiple of the 256 CUs x 2 resident workgroups: 3 (432 workgroups)
    # beats 4 (576) by 25 % on the FFN wgrads, 14 beats 8 on the 768x768 ones (microbench, profiles/r01_tn_split_sweep.json)
    TN_SPLIT_CANDIDATES = (0, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16)
    TN_VARIANT_CANDIDATES = (2, 26)     # 26 = LDS-DMA kernel + XCD-aware tile order + split-major block order
    # bench.py's live roofline: an event pair around a launch costs ~2 x 2.5 us of serialisation (5-6 % of the step when all 103
    # NT launches of a step are bracketed), so every 8th launch is sampled; 103 is coprime to 8, so successive steps sample
    # different launch positions and K >= 8 steps cover every launch of the step.
    PROF_EVERY = 8
    _nt_choice = {}          # shared across engines of one process: (M, N, K) -> variant
    _tn_choice = {}          # (M, N, K) -> splits

    def __init__(self, model):
        self._model = weakref.ref(model)
        self.packed = False
        self.gen = 0
        self.step_seed = 0
        self.base_seed = 0x5EED
        self.grads_dirty = False          # False -> next backward overwrites (beta = 0), True -> accumulates
        self.grad_ready_hook = None       # callable(bucket_index) set by the DDP wrapper
        self.post_backward_hook = None    # callable() set by the DDP wrapper
        self._ws = {}
        self._shadow = None
        self.prof = None                  # list -> every PROF_EVERY-th NT-GEMM launch is bracketed by HIP events (bench.py roofline)
        self._opt_stream = None           # optimizer stream of the pipelined FusedAdam step (optimization_fp16._step_pipelined)
        self._param_events = None         # {"nodecay" | bucket index: event} of the last pipelined optimizer step, consumed by forward
        self._params_done = None          # event behind its last chunk
        self.shard_plan = None            # vlp_amd.distributed.ShardPlan when the optimizer step is sharded over the ranks (VLP_DDP_MODE=sharded)
        self._param_works = None          # {"nodecay" | reducer bucket: collective work} of the last sharded step's parameter all-gather
        self.param_gather_stamps = None   # list -> (before, after) event pairs around every wait for a parameter all-gather (comm profile)
        self._side = None                 # second HIP stream for the layer wgrads (created on first use)
        self._side_busy = False           # True while backward may have work queued on it
        self._prof_ctr = 0
        self.varlen = self.VARLEN         # padding-free (packed) training step: True | False | "auto", see VARLEN
        self.varlen_readback_budget = 0   # auto mode: how many fresh dense masks may still be reduced + read back (pool warm-up)
        self._rb_streak = 0               # consecutive read-backs without a cache hit in between
        self._rb_warned = False
        self._pk_stage = None             # ring of pinned row_off staging buffers (+ events) for _packing
        self._pk_cache = {}               # kept-length tuple -> (row_off, row_map device tensors, M'): batches repeat in bench / epochs
        self._pk_lens = {}                # id(mask tensor) -> (weakref, version, ..., lens): lengths derived from a dense mask, once per tensor
        self.last_packed_rows = None      # M' of the latest packed forward (None: dense) -- bench.py / tests read it

    # ------------------------------------------------------------------------------------------
    # parameter packing
    # ------------------------------------------------------------------------------------------
    def _ordered_names(self, model):
        cfg = model.config
        tasks = model.tasks
        names = {n for n, _ in model.named_parameters()}
        decay, buckets = [], []

        def add(group):
            start = len(decay)
            decay.extend(group)
            buckets.append((start, len(decay)))

        head_unused = []
        if tasks == "vqa2":
            add(["ans_classifier.2.weight", "ans_classifier.0.weight"])
            head_unused.append("cls.predictions.transform.dense.weight")
        else:
            add(["cls.predictions.transform.dense.weight"])
        for i in reversed(range(cfg.num_hidden_layers)):
            L = "bert.encoder.layer.%d." % i
            add([L + "output.dense.weight", L + "intermediate.dense.weight", L + "attention.output.dense.weight",
                 L + "attention.self.query.weight", L + "attention.self.key.weight", L + "attention.self.value.weight"])
        # two slices: the embedding tables are final after embed_bwd (45 MB: the tied word embedding), the region projections only after
        # the three wgrads that follow -- the big slice's all-reduce overlaps them instead of waiting for the end of backward
        add(["bert.embeddings.position_embeddings.weight", "bert.embeddings.token_type_embeddings.weight",
             "bert.embeddings.word_embeddings.weight"])
        add(["vis_pe_embed.0.weight", "vis_embed.2.weight", "vis_embed.0.weight", "bert.pooler.dense.weight"] + head_unused)
        nodecay = sorted(n for n in names if is_no_decay(n))
        # q/k/v biases of a layer must be contiguous (packed QKV GEMM)
        for i in range(cfg.num_hidden_layers):
            L = "bert.encoder.layer.%d.attention.self." % i
            for n in (L + "query.bias", L + "key.bias", L + "value.bias"):
                nodecay.remove(n)
            nodecay.extend([L + "query.bias", L + "key.bias", L + "value.bias"])
        missing = names - set(decay) - set(nodecay)
        extra = (set(decay) | set(nodecay)) - names
        if missing or extra:
            raise RuntimeError("vlp_amd.Engine: unexpected parameter set (missing %s, unknown %s)" % (sorted(missing), sorted(extra)))
        return decay, nodecay, buckets

    def plan_layout(self, model=None):
        """Pure function of the parameter shapes (runs on CPU, no device needed): names per flat buffer, element offsets (every
        parameter on a 128-byte boundary), buffer sizes, and the gradient buckets = contiguous slices of the decay buffer in
        backward-completion order (head, layer N-1 .. 0, embedding tables, region projections)."""
        model = model if model is not None else self._model()
        numel = {n: p.numel() for n, p in model.named_parameters()}
        decay, nodecay, bucket_idx = self._ordered_names(model)
        names = {"decay": decay, "nodecay": nodecay}
        offsets, sizes = {}, {}
        for grp, ns in names.items():
            off, offs = 0, {}
            for n in ns:
                offs[n] = off
                off += _ru(numel[n], ALIGN)
            offsets[grp], sizes[grp] = offs, _ru(off, 8)
        buckets = []
        for s_, e_ in bucket_idx:
            lo = offsets["decay"][decay[s_]]
            hi = offsets["decay"][decay[e_ - 1]] + _ru(numel[decay[e_ - 1]], ALIGN)
            buckets.append((lo, min(hi, sizes["decay"])))
        return {"names": names, "offsets": offsets, "sizes": sizes, "buckets": buckets}

    def pack(self):
        """Move every parameter into the flat buffers (idempotent).  Needs fp16 parameters on a GPU:
        this is the `model.half(); model.to(device)` state of run_img2txt_dist.py:370-377."""
        if self.packed:
            return
        model = self._model()
        params = dict(model.named_parameters())
        dev = next(iter(params.values())).device
        if dev.type != "cuda":
            raise RuntimeError("vlp_amd: the model must be on an MI355X (`model.to('cuda')`); there is no CPU path")
        for n, p in params.items():
            if p.dtype != torch.float16:
                raise RuntimeError("vlp_amd: parameters must be fp16 (`model.half()`, i.e. --fp16): %s is %s. "
                                   "The fp32 path of the reference is not implemented (DESIGN.md, out of scope)." % (n, p.dtype))
        K.load()
        lay = self.plan_layout(model)
        self.names, self.offsets, self.sizes, self.buckets = lay["names"], lay["offsets"], lay["sizes"], lay["buckets"]
        self.flat, self.gflat = {}, {}
        for grp, names in self.names.items():
            offs, total = self.offsets[grp], self.sizes[grp]
            flat = torch.zeros(total, device=dev, dtype=torch.float16)
            gflat = torch.zeros(total, device=dev, dtype=torch.float16)
            for n in names:
                p = params[n]
                view = flat[offs[n]:offs[n] + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = gflat[offs[n]:offs[n] + p.numel()].view(p.shape)
                p._vlp_engine = self
            self.flat[grp], self.gflat[grp] = flat, gflat
        self._params = params
        self.device = dev
        self.packed = True
        self.grads_dirty = False
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._unused = set(["bert.pooler.dense.weight", "bert.pooler.dense.bias"])
        if model.tasks == "vqa2":
            self._unused |= {"cls.predictions.bias", "cls.predictions.transform.dense.weight", "cls.predictions.transform.dense.bias",
                             "cls.predictions.transform.LayerNorm.weight", "cls.predictions.transform.LayerNorm.bias"}

    def invalidate(self):
        """Called when nn.Module._apply() replaced parameter storage (.half()/.to()/.cpu())."""
        self.packed = False
        self._ws = {}
        self._shadow = None
        self._ln_tab = None

    def P(self, name):
        return self._params[name].data

    def G(self, name):
        return self._params[name].grad

    def unused_parameter_names(self):
        """Parameters that receive no gradient (static per task).  The pooler is used by the vis_pretext branch only:
        it leaves the set while the latest forward ran with mask_image_regions."""
        if getattr(self, "_pretext_on", False):
            return set(self._unused) - {"bert.pooler.dense.weight", "bert.pooler.dense.bias"}
        return set(self._unused)

    # ------------------------------------------------------------------------------------------
    # pipelined optimizer step: parameter chunks become valid one by one (events on the optimizer stream)
    # ------------------------------------------------------------------------------------------
    def optimizer_stream(self):
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=self.device)
        return self._opt_stream

    def set_param_events(self, events, done):
        self._param_events, self._params_done = events, done

    def wait_params(self, key=None, host=False, consume=True):
        """Make the current stream (host=True: the calling thread) wait until the optimizer stream has written parameter chunk `key`
        ("nodecay" or a bucket index); key=None: all of them (also orders the optimizer's reads of the gradient buffers before later work)."""
        if self._param_works is not None:          # sharded step: parameter chunks arrive by all-gather, one collective per bucket
            works = self._param_works
            prof = self.param_gather_stamps            # bench.py comm profile: how long the forward stood still for its parameters
            if prof is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            if key is None:
                for w in works.values():
                    if w is not None:
                        w.wait()                   # (RCCL: the current stream waits; no host sync)
                self._param_works = None
            else:
                k = key if key == "nodecay" else self.shard_plan.bucket_of_slice[key]
                w = works.get(k)
                if w is not None:
                    w.wait()
                    if consume:                    # (consume=False: a wait on ANOTHER stream -- the main stream still has to wait for this bucket itself)
                        works[k] = None
            if prof is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                prof.append((e0, e1))
        if self._params_done is None:
            return
        if key is None:
            if host:
                self._params_done.synchronize()
            else:
                torch.cuda.current_stream().wait_event(self._params_done)
            if host:
                self._param_events, self._params_done = None, None
            return
        ev = self._param_events.get(key) if self._param_events else None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def zero_placeholder(self, device):
        """The shared read-only [1] fp32 zero that stands for a loss the task does not have (modeling.py:1096-1098, 1133)."""
        z = getattr(self, "_zero1", None)
        if z is None or z.device != device:
            z = self._zero1 = torch.zeros(1, device=device, dtype=torch.float32)
        return z

    def is_zero_placeholder(self, t):
        """True when `t` is the shared zero a forward hands out for a loss the task does not have (train loops skip it instead of
        launching `+ 0`).  The placeholder is shared by every forward of this engine: do not modify it in place."""
        return t is getattr(self, "_zero1", None)

    def zero_grad(self):
        """optimizer.zero_grad() of the train loop (run_img2txt_dist.py:585): no memset -- the next
        backward simply overwrites (beta = 0) instead of accumulating."""
        self.grads_dirty = False

    # ------------------------------------------------------------------------------------------
    # workspaces
    # ------------------------------------------------------------------------------------------
    def _workspace(self, B, L, P):
        key = (B, L, P)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        model = self._model()
        cfg = model.config
        H, I, A, NL, Nv, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_hidden_layers, model.len_vis_input, cfg.vocab_size
        M, Mv = B * L, B * Nv
        dev = self.device

        def h(*s):
            return torch.empty(*s, device=dev, dtype=torch.float16)

        def f(*s):
            return torch.empty(*s, device=dev, dtype=torch.float32)

        ws = {"Lp": _ru(L, 32)}
        ws["maskb"] = torch.empty(B, L, ws["Lp"], device=dev, dtype=torch.uint8)
        ws["maskt"] = torch.empty(B, ws["Lp"], ws["Lp"], device=dev, dtype=torch.uint8)
        ws["img16"], ws["vpe_in"], ws["wpe_pad"] = h(Mv, 2048), h(Mv, PE_PAD), h(H, PE_PAD)
        ws["h1"], ws["vis_h"], ws["vispe_h"] = h(Mv, 2048), h(Mv, H), h(Mv, H)
        ws["emb_pre"], ws["x0"] = h(M, H), h(M, H)
        ws["stat0"] = (f(M), f(M))
        lay = []
        for _ in range(NL):
            lay.append({"qkv": h(M, 3 * H), "ctx": h(M, H), "lse": f(B, A, L), "pre1": h(M, H), "x1": h(M, H),
                        "st1": (f(M), f(M)), "z": h(M, I), "g": h(M, I), "pre2": h(M, H), "x2": h(M, H), "st2": (f(M), f(M))})
        ws["layers"] = lay
        # heads
        Vp = _ru(V, 64)
        ws["Vp"] = Vp
        if P > 0:
            R = B * P
            ws.update(sel=h(R, H), tz=h(R, H), tg=h(R, H), tln=h(R, H), tstat=(f(R), f(R)), logits=h(R, Vp), dlogits=h(R, Vp),
                      dlT=h(Vp, _ru(R, 64)),
                      lse_ce=f(R), coef=f(R), row_loss=f(R), dtln=h(R, H), dtg=h(R, H), dtz=h(R, H), dsel=h(R, H))
        ws["loss"] = f(260)
        if model.tasks == "vqa2":
            NA = model.num_answers
            NAp = _ru(NA, 64)
            ws.update(NAp=NAp, vq_e=h(B, H), vq_a1=h(B, 2 * H), vq_logits=h(B, NAp), vq_dlogits=h(B, NAp), vq_dz1=h(B, 2 * H), vq_de=h(B, H))
        # backward scratch (shared by all layers)
        ws.update(dx=h(M, H), dx_alt=h(M, H), dpre=h(M, H), dpre_d=h(M, H), dz=h(M, I), dctx=h(M, H), dqkv=h(M, 3 * H),
                  delta=f(B, A, L), d_vis_h=h(Mv, H), d_vispe_h=h(Mv, H), dz1v=h(Mv, 2048), dwpe_pad=h(H, PE_PAD), acc32=f(K.embed_bwd_workspace_floats(B, L, Nv, H)))
        # dY operands of the weight-gradient GEMMs, double-buffered by layer parity: the wgrads of layer i run on a side stream
        # while the main stream already works on layer i-1
        ws["dyset"] = [{"dpre2": h(M, H), "dpre2_d": h(M, H), "dpre1": h(M, H), "dpre1_d": h(M, H), "dz": h(M, I), "dqkv": h(M, 3 * H)}
                       for _ in range(2)]
        tn_bytes = max(K.gemm_tn_workspace_bytes(M, I, H), K.gemm_tn_workspace_bytes(Mv, 2048, 2048), K.gemm_tn_workspace_bytes(M, 3 * H, H))
        ws["tn_ws"] = torch.empty(tn_bytes, device=dev, dtype=torch.uint8)
        ws["tn_ws_main"] = torch.empty(max(K.gemm_tn_workspace_bytes(V, _ru(B * max(P, 1), 64), H), 1 << 20), device=dev, dtype=torch.uint8)
        ws["cs_ws"] = torch.empty(max(K.colsum_workspace_bytes(M, I), K.colsum_workspace_bytes(B * max(P, 1), Vp)), device=dev, dtype=torch.uint8)
        ws["ln_ws"] = torch.empty(K.layernorm_bwd_workspace_bytes(max(H, 8)), device=dev, dtype=torch.uint8)
        # one private partials slot per encoder / embedding LayerNorm: their dgamma / dbeta second stages run as ONE launch at the
        # end of backward (vlp_layernorm_bwd_reduce_batched) instead of 2 * layers + 1 tiny reduce kernels
        ws["ln_slot_bytes"] = K.layernorm_bwd_workspace_bytes(max(H, 8))
        ws["ln_slots"] = torch.empty((2 * NL + 1) * ws["ln_slot_bytes"], device=dev, dtype=torch.uint8)
        self._ws[key] = ws
        return ws

    def _shadows(self):
        """Transposed weight copies W^T (zero padded so that every dgrad K is a multiple of 64)."""
        if self._shadow is not None:
            return self._shadow
        model = self._model()
        cfg = model.config
        H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
        dev = self.device

        def h(*s):
            return torch.zeros(*s, device=dev, dtype=torch.float16)

        sh = {"layers": [{"qkvT": h(H, 3 * H), "oT": h(H, H), "w1T": h(H, I), "w2T": h(I, H)} for _ in range(cfg.num_hidden_layers)],
              "v2T": h(2048, H)}
        if model.tasks == "vqa2":
            NAp = _ru(model.num_answers, 64)
            sh.update(a2T=h(2 * H, NAp), a0T=h(H, 2 * H))
        else:
            sh.update(tT=h(H, H))
        self._shadow = sh
        return sh

    def _ln_table(self):
        """Device table [2 * layers + 1, 2] of (dgamma, dbeta) addresses: slot 2i = attention.output.LayerNorm of layer i, 2i + 1 =
        output.LayerNorm of layer i, last = embeddings.LayerNorm (addresses of the flat gradient buffer are stable)."""
        if getattr(self, "_ln_tab", None) is None:
            NL = self._model().config.num_hidden_layers
            rows = []
            for i in range(NL):
                L = "bert.encoder.layer.%d." % i
                for n in ("attention.output.LayerNorm", "output.LayerNorm"):
                    rows.append([self.G(L + n + ".weight").data_ptr(), self.G(L + n + ".bias").data_ptr()])
            rows.append([self.G("bert.embeddings.LayerNorm.weight").data_ptr(), self.G("bert.embeddings.LayerNorm.bias").data_ptr()])
            self._ln_tab = torch.tensor(rows, dtype=torch.int64, device=self.device)
        return self._ln_tab

    def _refresh_shadows(self):
        """All W^T shadows of the step in one batched launch (descriptor table cached: buffer addresses are stable)."""
        sh = self._shadows()
        if "batch" not in sh:
            model = self._model()
            cfg = model.config
            H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
            items = []
            for i, s in enumerate(sh["layers"]):
                L = "bert.encoder.layer.%d." % i
                items.append((self.P(L + "attention.self.query.weight"), H, s["qkvT"], 3 * H, 3 * H, H, 3 * H))   # packed [3H, H] -> [H, 3H]
                items.append((self.P(L + "attention.output.dense.weight"), H, s["oT"], H, H, H, H))
                items.append((self.P(L + "intermediate.dense.weight"), H, s["w1T"], I, I, H, I))
                items.append((self.P(L + "output.dense.weight"), I, s["w2T"], H, H, I, H))
            items.append((self.P("vis_embed.2.weight"), 2048, sh["v2T"], H, H, 2048, H))
            if model.tasks == "vqa2":
                NA = model.num_answers
                items.append((self.P("ans_classifier.2.weight"), 2 * H, sh["a2T"], _ru(NA, 64), NA, 2 * H, _ru(NA, 64)))
                items.append((self.P("ans_classifier.0.weight"), H, sh["a0T"], 2 * H, 2 * H, H, 2 * H))
            else:
                items.append((self.P("cls.predictions.transform.dense.weight"), H, sh["tT"], H, H, H, H))
            sh["batch"] = K.make_transpose_batch(items, self.device)
        K.transpose_batched(sh["batch"])

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def _nt_variant(self, x, w, y, M, N, Kd, kw):
        """Staging/tiling variant for this problem size.  Deterministic (vlp_amd.tuning: committed table, else a shape
        heuristic) so that every box runs the same kernels and produces the same bits; with VLP_AUTOTUNE=1 every candidate
        (all compute the same contraction) is timed on the real call the first time a shape is seen."""
        if self.GEMM_NT_VARIANT is not None:
            return self.GEMM_NT_VARIANT
        key = (M, N, Kd)
        v = Engine._nt_choice.get(key)
        if v is not None:
            return v
        if not tuning.AUTOTUNE:
            v = Engine._nt_choice[key] = tuning.nt_variant(M, N, Kd)
            return v
        best, best_t = self.NT_CANDIDATES[0], float("inf")
        cands = self.NT_CANDIDATES_SKINNY if M <= 1024 else self.NT_CANDIDATES
        if M * N >= 128 * 128 * 4:          # tiny problems: not worth timing
            torch.cuda.synchronize()        # nothing else (e.g. side-stream wgrads) may run while candidates are timed
            for rnd in range(2):              # two interleaved rounds, best-of: robust against clock / neighbour noise
                for cand in cands:
                    if (cand & 7 == 5 and cand < 64 or cand == 73) and N < 1024:
                        continue                  # 256-wide n tiles leave most CUs idle on narrow outputs
                    K.gemm_nt(x, w, y, M, N, Kd, variant=cand, **kw)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(4 if M > 1024 else 12):
                        K.gemm_nt(x, w, y, M, N, Kd, variant=cand, **kw)
                    e1.record()
                    e1.synchronize()
                    t = e0.elapsed_time(e1)
                    if t < best_t:
                        best, best_t = cand, t
        Engine._nt_choice[key] = best
        tuning.remember("nt", M, N, Kd, best)
        return best

    def _nt(self, x, w, y, M, N, Kd, **kw):
        v = self._nt_variant(x, w, y, M, N, Kd, kw)
        if self.prof is not None:
            self._prof_ctr += 1
        if self.prof is None or self._prof_ctr % self.PROF_EVERY:
            K.gemm_nt(x, w, y, M, N, Kd, variant=v, **kw)
            return
        # events are recorded on torch's current stream == the stream handed to the C ABI.  While backward has weight-gradient GEMMs
        # in flight on the side stream, a sampled launch waits for them (and they for it), so that the bracket times ONE kernel.
        side = self._side if self._side_busy else None
        main = torch.cuda.current_stream()
        if side is not None:
            main.wait_stream(side)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.gemm_nt(x, w, y, M, N, Kd, variant=v, **kw)
        e1.record()
        if side is not None:
            side.wait_stream(main)
        self.prof.append((e0, e1, 2.0 * M * N * Kd))

    SKINNY_SPLITS = (2, 3, 4, 6, 8, 12, 16)
    _skinny_choice = {}      # (M, N, K) -> ("v", variant) | ("s", splits)

    def _nt_skinny(self, x, w, y, M, N, Kd, skws, tune_ws=None, **kw):
        """NT GEMM of the incremental decoder (M = sequences x 2 rows): the ordinary kernels have only N/128 workgroups to run, so the
        split-K form (vlp_gemm_nt_splitk) and every ordinary variant are timed once per shape.  `tune_ws()` returns same-shaped weight
        tensors (the same projection in all layers) to cycle through while timing: in a real token step every weight is read once, i.e.
        from HBM / Infinity Cache, not from a warm L2 -- timing one weight back to back would rank the candidates for the wrong regime."""
        if M > 1024 or N > 4096 or Kd < 256:
            return self._nt(x, w, y, M, N, Kd, **kw)
        key = (M, N, Kd)
        ch = Engine._skinny_choice.get(key)
        if ch is None and not tuning.AUTOTUNE:
            ch = Engine._skinny_choice[key] = tuning.skinny_choice(M, N, Kd)
        if ch is None:
            wl = list(tune_ws()) if tune_ws is not None else [w]
            reps = max(12, len(wl))

            def timed(fn):
                fn(wl[0])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    fn(wl[i % len(wl)])
                e1.record()
                e1.synchronize()
                return e0.elapsed_time(e1)
            torch.cuda.synchronize()
            cands = [("v", v) for v in self.NT_CANDIDATES_SKINNY] + [("s", sp) for sp in self.SKINNY_SPLITS if Kd // 64 >= sp]
            score = {c: float("inf") for c in cands}
            for rnd in range(3):                     # interleaved rounds, best-of per candidate: these are 5-40 us kernels, one noisy
                for c in cands:                      # sample would otherwise pin a bad choice for the life of the process
                    if c[0] == "v":
                        t = timed(lambda ww, v=c[1]: K.gemm_nt(x, ww, y, M, N, Kd, variant=v, **kw))
                    else:
                        t = timed(lambda ww, sp=c[1]: K.gemm_nt_splitk(x, ww, y, M, N, Kd, sp, skws, **kw))
                    score[c] = min(score[c], t)
            ch = min(cands, key=lambda c: score[c])
            K.gemm_nt(x, w, y, M, N, Kd, variant=1, **kw)        # leave y as computed from the caller's weight
            Engine._skinny_choice[key] = ch
            tuning.remember("sk", M, N, Kd, ch)
        if ch[0] == "v":
            K.gemm_nt(x, w, y, M, N, Kd, variant=ch[1], **kw)
        else:
            K.gemm_nt_splitk(x, w, y, M, N, Kd, ch[1], skws, **kw)

    def _pretext_workspace(self, ws, B, Pm):
        """Buffers of the mask_image_regions / vis_pretext branch (modeling.py:1049-1056, 1113-1131), created on first use."""
        key = ("pt", Pm)
        if key not in ws:
            model = self._model()
            H, Nv, dev = model.config.hidden_size, model.len_vis_input, self.device
            ws[key] = dict(rmask=torch.empty(B * Nv, device=dev, dtype=torch.uint8), pos0=torch.zeros(B, 1, device=dev, dtype=torch.long),
                           sel0=torch.empty(B, H, device=dev, dtype=torch.float16), pooled=torch.empty(B, H, device=dev, dtype=torch.float16),
                           probs=torch.empty(B, Pm, Pm, device=dev, dtype=torch.float32), sample=torch.empty(B, device=dev, dtype=torch.float32),
                           loss=torch.empty(1, device=dev, dtype=torch.float32), dpool=torch.empty(B, H, device=dev, dtype=torch.float16),
                           dsel0=torch.empty(B, H, device=dev, dtype=torch.float16), pT=torch.empty(H, H, device=dev, dtype=torch.float16))
        return ws[key]

    def _kept_lengths(self, attention_mask, masked_pos, B, L, Nv):
        """Per-sample number of leading positions the packed step keeps (host ints): 1 + the last key column ANY query attends, and at
        least what the heads read (masked_pos, position Nv+1).  Every dropped position is therefore attended by no query at all.
        MaskSpec carries the lengths on the host (lens_host = second_end, the loader knows them); a dense int64 mask is reduced on the
        device and read back ONCE per tensor object (the result is remembered while that tensor is alive and unmodified -- a device-
        resident batch pool, bench.py / --synthetic, pays the read-back in its first pass only)."""
        if isinstance(attention_mask, MaskSpec):
            # (lens_host is the loader's promise that nothing past it is attended OR read: masked_pos lies inside the tokens by construction,
            # seq2seq_loader.py:256-265; the gather clamps a position past the kept rows to the last kept row, as the dense kernel clamps to L - 1)
            if attention_mask.lens_host is None:
                return None
            lens = [max(int(n), Nv + 2) for n in attention_mask.lens_host]
            return lens if len(lens) == B and max(lens) <= L else None
        if attention_mask is None or attention_mask.dim() != 3:
            return None
        key = id(attention_mask)
        hit = self._pk_lens.get(key)
        if hit is not None and hit[0]() is attention_mask and hit[1] == attention_mask._version and \
                (hit[2] is None) == (masked_pos is None) and (masked_pos is None or (hit[2]() is masked_pos and hit[4] == masked_pos._version)):
            self._rb_streak = 0
            return hit[3]           # the SAME tensor objects, unmodified (weak references: a recycled id() cannot alias)
        # a fresh dense mask: its lengths cost a reduction + a device -> host read-back in front of this forward
        if self.varlen == "auto":
            if self.varlen_readback_budget <= 0:
                self._warn_readback("the padding-free step is OFF for this batch: its dense [B, L, L] attention mask is a new tensor, and deriving the kept "
                                    "lengths from it would stall the launch queue once per step")
                return None
            self.varlen_readback_budget -= 1
        else:
            self._rb_streak += 1
            if self._rb_streak > 8:
                self._warn_readback("the padding-free step derives the kept lengths of a NEW dense [B, L, L] attention mask every step: one device -> host "
                                    "read-back (a launch-queue stall) per step")
        cols = (attention_mask != 0).any(dim=1)                                            # [B, L]: key column attended by some query
        idx = torch.arange(1, L + 1, device=cols.device, dtype=torch.int32)
        n = (cols.to(torch.int32) * idx).amax(dim=1)
        n = torch.clamp(n, min=Nv + 2)
        if masked_pos is not None and masked_pos.numel() > 0:
            n = torch.maximum(n, (masked_pos.to(torch.int32).amax(dim=1) + 1).clamp(max=L))
        lens = [int(v) for v in n.tolist()]                                              # the one host read-back
        if len(self._pk_lens) > 64:
            self._pk_lens = {k: v for k, v in self._pk_lens.items() if v[0]() is not None}
            if len(self._pk_lens) > 64:
                self._pk_lens.clear()
        self._pk_lens[key] = (weakref.ref(attention_mask), attention_mask._version, weakref.ref(masked_pos) if masked_pos is not None else None, lens,
                              masked_pos._version if masked_pos is not None else 0)
        return lens

    def _warn_readback(self, what):
        if self._rb_warned:
            return
        self._rb_warned = True
        import warnings
        warnings.warn("vlp_amd: " + what + ".  Hand the engine vlp_amd.input_prep.MaskSpec (three int32 per sample + the lengths on the host: what "
                      "vlp_amd.data.BatchPrefetcher delivers) in place of the dense mask, or keep the mask tensors resident and reuse them; "
                      "VLP_VARLEN=0 silences this, VLP_VARLEN=1 accepts the read-back.", VlpPerformanceWarning, stacklevel=3)

    PK_STAGE_SLOTS = 8

    def _packing(self, lens, B, L):
        """(row_off int32 [B+1], row_map int32 [M'], M') on the device for the kept lengths `lens`.  Batches of a resident pool / an epoch
        replay hit the cache; a real loader produces a new length tuple every step: its offsets go through a small RING of pinned staging
        buffers (no hipHostMalloc per step, nothing pinned kept per cache entry -- ADVICE r5), one async H2D copy + one rowmap_build launch."""
        key = (L,) + tuple(lens)
        ent = self._pk_cache.get(key)
        if ent is None:
            if self._pk_stage is None or self._pk_stage[0][0].numel() < B + 1:
                self._pk_stage = [[torch.empty(max(B + 1, 257), dtype=torch.int32).pin_memory(), torch.cuda.Event()] for _ in range(self.PK_STAGE_SLOTS)]
                self._pk_stage_i = 0
            host, ev = self._pk_stage[self._pk_stage_i]
            self._pk_stage_i = (self._pk_stage_i + 1) % self.PK_STAGE_SLOTS
            ev.synchronize()                       # the copy issued from this buffer PK_STAGE_SLOTS misses ago (long done; never-recorded: returns at once)
            off = np.zeros(B + 1, dtype=np.int32)
            np.cumsum(np.asarray(lens, dtype=np.int32), out=off[1:])
            host.numpy()[:B + 1] = off
            row_off = torch.empty(B + 1, dtype=torch.int32, device=self.device)
            row_off.copy_(host[:B + 1], non_blocking=True)
            ev.record()
            Mp = int(off[-1])
            row_map = torch.empty(Mp, dtype=torch.int32, device=self.device)
            K.rowmap_build(row_off, B, L, row_map)
            if len(self._pk_cache) >= 64:
                self._pk_cache.clear()
            ent = self._pk_cache[key] = (row_off, row_map, Mp)
        return ent[0], ent[1], ent[2]

    def forward(self, vis_feats, vis_pe, input_ids, token_type_ids, attention_mask, masked_pos, train, want_mlm, want_vqa,
                vis_masked_pos=None):
        """Runs embeddings + encoder (+ heads' forward up to the logits).  Returns the _State.  vis_masked_pos ([B, Pm] int64, values
        1..Nv): the mask_image_regions branch -- those region rows enter the encoder as zeros and the pooled output is computed
        for the pretext loss (pretext_loss())."""
        self.pack()
        model = self._model()
        cfg = model.config
        H, I, A, NL, Nv, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_hidden_layers, model.len_vis_input, cfg.vocab_size
        B, L = input_ids.shape
        if H != A * 64:
            raise RuntimeError("vlp_amd: attention kernels need head_dim == 64 (hidden %d, heads %d)" % (H, A))
        if vis_feats.shape[1] != Nv or vis_feats.shape[2] != 2048 or vis_pe.shape[2] != PE_DIM:
            raise RuntimeError("vlp_amd: expected vis_feats [B,%d,2048] and vis_pe [B,%d,%d]" % (Nv, Nv, PE_DIM))
        raw_regions = isinstance(vis_pe, RawRegions)
        mask_spec = isinstance(attention_mask, MaskSpec)
        if L < Nv + 2:
            raise RuntimeError("vlp_amd: sequence length %d too short for %d regions" % (L, Nv))
        P = masked_pos.shape[1] if (want_mlm and masked_pos is not None and masked_pos.numel() > 0) else 0
        ws = self._workspace(B, L, P)
        self.gen += 1
        st = _State()
        st.gen, st.B, st.L, st.P, st.ws = self.gen, B, L, P, ws
        p = cfg.hidden_dropout_prob if train else 0.0
        pa = cfg.attention_probs_dropout_prob if train else 0.0
        st.p_drop = (p, pa)
        if train and (p > 0 or pa > 0):
            self.step_seed += 1
        seed = st.seed = self.base_seed + self.step_seed
        M, Mv = B * L, B * Nv
        # ---- padding-free layout (opt-in): row_off / row_map / M' from the mask's own kept lengths --------------------------------
        ro = rm = None
        if self.varlen and (train or torch.is_grad_enabled()) and L <= 192 and os.environ.get("VLP_ATTN_BWD") is None:
            lens = self._kept_lengths(attention_mask, masked_pos if P > 0 else None, B, L, Nv)
            if lens is not None and sum(lens) < M:
                ro, rm, M = self._packing(lens, B, L)
        st.pk = (ro, rm, M) if ro is not None else None
        self.last_packed_rows = M if ro is not None else None

        # ---- inputs -----------------------------------------------------------------------------
        want_t = ws["maskt"] if train or torch.is_grad_enabled() else None
        pt = None
        if vis_masked_pos is not None and vis_masked_pos.numel() > 0:
            Pm = vis_masked_pos.shape[1]
            pt = self._pretext_workspace(ws, B, Pm)
            st_vmp = vis_masked_pos.to(torch.long).contiguous()
            K.region_mask_build(st_vmp, pt["rmask"], B, Pm, Nv)
        self._pretext_on = pt is not None
        st.pretext = (pt, st_vmp) if pt is not None else None
        vf = vis_feats.reshape(Mv, 2048)
        if vf.dtype == torch.float32:
            K.copy2d(vf.contiguous(), 2048, True, ws["img16"], 2048, Mv, 2048, 2048)
            img = ws["img16"]
        else:
            img = vf.contiguous()
        st.batch = (img, input_ids.contiguous(), token_type_ids.contiguous(), masked_pos)

        def prep_pe():
            # K-padded box / class encoding (operand of the third region GEMM) and the K-padded copy of its weight
            if raw_regions:                     # raw boxes + class probabilities -> K-padded encoding (seq2seq_loader.py:338-351)
                vis_pe.check(B, Nv)
                K.vis_pe_prep(vis_pe.bbox, vis_pe.cls_prob.reshape(Mv, PE_DIM - 6), ws["vpe_in"], B, Nv, PE_DIM - 6, PE_PAD)
            else:
                vp = vis_pe.reshape(Mv, PE_DIM).contiguous()
                K.copy2d(vp, PE_DIM, vp.dtype == torch.float32, ws["vpe_in"], PE_PAD, Mv, PE_DIM, PE_PAD)
            # parameters written by a pipelined optimizer step become readable chunk by chunk (wait_params is a no-op otherwise)
            # (reads a parameter of the embeddings bucket: AFTER the wait, or a pipelined step would project with last step's weight)
            self.wait_params(len(self.buckets) - 1, consume=False)      # region projections (this may run on the side stream)
            K.copy2d(self.P("vis_pe_embed.0.weight"), PE_DIM, False, ws["wpe_pad"], PE_PAD, H, PE_DIM, PE_PAD)

        def prep_mask(am):
            if mask_spec:                       # per-sample lengths -> packed masks on the device (seq2seq_loader.py:292-301)
                am.check(B, L)
                # (BLOCK_MASKED_REGION_KEYS: opt-in, the reference's own loader leaves the masked regions' key columns attendable, see above)
                K.mask_build(am.second_st, am.second_end, am.is_s2s, ws["maskb"], B, L, ws["Lp"], out_t=want_t,
                             region_mask=pt["rmask"] if (pt is not None and self.BLOCK_MASKED_REGION_KEYS) else None, Nv=Nv)
            else:
                if am is None:
                    am = torch.ones(B, L, dtype=torch.long, device=input_ids.device)
                if am.dim() == 2:       # modeling.py:818-819
                    am = am[:, None, :].expand(B, L, L)
                am = am.to(torch.long).contiguous()
                K.mask_pack(am, ws["maskb"], B, L, ws["Lp"], out_t=want_t)

        # ---- side stream underneath the forward (training): (1) the K-padded box / class operand of the third region GEMM, (2) the packed
        # attention masks (first read by layer 0's attention, ~190 us into the forward), (3) the W^T shadows of this step's dgrad GEMMs (the
        # weights are final once the optimizer has stepped; 78 us, first read at the head of backward).  The main stream starts the region
        # projections at once and waits for (1) / (2) where it first reads them.
        self._shadow_ev = None
        pe_ev = mask_ev = None
        if (train or torch.is_grad_enabled()) and self.WGRAD_SIDE_STREAM and self.SHADOW_ON_SIDE:
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
                self._side_done = [None, None]
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                prep_pe()
                pe_ev = torch.cuda.Event()
                pe_ev.record(self._side)
                prep_mask(attention_mask)
                mask_ev = torch.cuda.Event()
                mask_ev.record(self._side)
                if self._params_done is not None:
                    self._side.wait_event(self._params_done)        # the transposes read every weight matrix
                if self._param_works is not None:
                    # sharded optimizer step: the parameters arrive by per-bucket all-gathers that may still be in flight; the
                    # transposes read EVERY weight matrix, so this stream waits for all of them (the main stream keeps waiting per
                    # bucket where the forward first reads it -- Work.wait() may be called again there)
                    for w in self._param_works.values():
                        if w is not None:
                            w.wait()
                self._refresh_shadows()
                self._shadow_ev = torch.cuda.Event()
                self._shadow_ev.record(self._side)
        else:
            prep_mask(attention_mask)
            prep_pe()

        self.wait_params("nodecay")
        self.wait_params(len(self.buckets) - 1)             # region projections
        self.wait_params(len(self.buckets) - 2)             # embedding tables
        # ---- region projections (modeling.py:1003-1018,1035-1036) ---------------------------------
        self._nt(img, self.P("vis_embed.0.weight"), ws["h1"], Mv, 2048, 2048, bias=self.P("vis_embed.0.bias"), act=K.ACT_RELU)
        self._nt(ws["h1"], self.P("vis_embed.2.weight"), ws["vis_h"], Mv, H, 2048, bias=self.P("vis_embed.2.bias"), act=K.ACT_RELU,
                 dropout_p=p, seed=seed, rng_stream=1001)
        if pe_ev is not None:
            torch.cuda.current_stream().wait_event(pe_ev)
        self._nt(ws["vpe_in"], ws["wpe_pad"], ws["vispe_h"], Mv, H, PE_PAD, bias=self.P("vis_pe_embed.0.bias"), act=K.ACT_RELU,
                 dropout_p=p, seed=seed, rng_stream=1002)
        # ---- embeddings (modeling.py:217-241) ------------------------------------------------------
        E = "bert.embeddings."
        K.embed_fwd(st.batch[1], st.batch[2], self.P(E + "word_embeddings.weight"), self.P(E + "position_embeddings.weight"),
                    self.P(E + "token_type_embeddings.weight"), ws["vis_h"], ws["vispe_h"], ws["emb_pre"], B, L, Nv, H,
                    region_mask=pt["rmask"] if pt is not None else None, row_map=rm, rows=M if rm is not None else 0)
        K.layernorm_fwd(ws["emb_pre"], self.P(E + "LayerNorm.weight"), self.P(E + "LayerNorm.bias"), ws["x0"], M, H, ws["stat0"][0], ws["stat0"][1],
                        dropout_p=p, seed=seed, rng_stream=1000, row_map=rm)
        # ---- encoder (modeling.py:268-372) -----------------------------------------------------------
        x = ws["x0"]
        scale = 1.0 / math.sqrt(H // A)
        for i in range(NL):
            Ln = "bert.encoder.layer.%d." % i
            a = ws["layers"][i]
            self.wait_params(NL - i)                            # bucket of layer i
            self._nt(x, self.P(Ln + "attention.self.query.weight"), a["qkv"], M, 3 * H, H, bias=self.P(Ln + "attention.self.query.bias"))
            if i == 0 and mask_ev is not None:
                torch.cuda.current_stream().wait_event(mask_ev)
            K.attn_fwd(a["qkv"], ws["maskb"], a["ctx"], a["lse"], B, L, A, scale, dropout_p=pa, seed=seed, rng_stream=16 * i + 1, row_off=ro)
            self._nt(a["ctx"], self.P(Ln + "attention.output.dense.weight"), a["pre1"], M, H, H, bias=self.P(Ln + "attention.output.dense.bias"),
                     residual=x, dropout_p=p, seed=seed, rng_stream=16 * i + 2, row_map=rm)
            K.layernorm_fwd(a["pre1"], self.P(Ln + "attention.output.LayerNorm.weight"), self.P(Ln + "attention.output.LayerNorm.bias"),
                            a["x1"], M, H, a["st1"][0], a["st1"][1])
            # a["z"] receives gelu'(z), not z: the only consumer is the FFN-down dgrad epilogue, which then multiplies by a stored
            # number instead of re-evaluating erf + exp per element
            self._nt(a["x1"], self.P(Ln + "intermediate.dense.weight"), a["g"], M, I, H, bias=self.P(Ln + "intermediate.dense.bias"),
                     preact=a["z"], act=K.ACT_GELU_SAVE_GRAD)
            self._nt(a["g"], self.P(Ln + "output.dense.weight"), a["pre2"], M, H, I, bias=self.P(Ln + "output.dense.bias"),
                     residual=a["x1"], dropout_p=p, seed=seed, rng_stream=16 * i + 3, row_map=rm)
            K.layernorm_fwd(a["pre2"], self.P(Ln + "output.LayerNorm.weight"), self.P(Ln + "output.LayerNorm.bias"), a["x2"], M, H,
                            a["st2"][0], a["st2"][1])
            x = a["x2"]
        # ---- heads ------------------------------------------------------------------------------------
        self.wait_params(0)
        st.has_mlm = P > 0
        if P > 0:
            C = "cls.predictions."
            R = B * P
            K.gather_rows(x, H, masked_pos.contiguous(), ws["sel"], H, B, P, L, H, row_off=ro)
            self._nt(ws["sel"], self.P(C + "transform.dense.weight"), ws["tg"], R, H, H, bias=self.P(C + "transform.dense.bias"),
                     preact=ws["tz"], act=K.ACT_GELU)
            K.layernorm_fwd(ws["tg"], self.P(C + "transform.LayerNorm.weight"), self.P(C + "transform.LayerNorm.bias"), ws["tln"], R, H,
                            ws["tstat"][0], ws["tstat"][1])
            self._nt(ws["tln"], self.P("bert.embeddings.word_embeddings.weight"), ws["logits"], R, V, H, bias=self.P(C + "bias"), ldy=ws["Vp"])
        if want_vqa:
            NA = model.num_answers
            K.vqa_mul_fwd(x, ws["vq_e"], B, L, Nv, H, row_off=ro)
            self._nt(ws["vq_e"], self.P("ans_classifier.0.weight"), ws["vq_a1"], B, 2 * H, H, bias=self.P("ans_classifier.0.bias"), act=K.ACT_RELU)
            self._nt(ws["vq_a1"], self.P("ans_classifier.2.weight"), ws["vq_logits"], B, NA, 2 * H, bias=self.P("ans_classifier.2.bias"), ldy=ws["NAp"])
        if pt is not None:
            # BertPooler (modeling.py:411-417) -- only this branch reads it -- and the pretext loss (:1113-1131)
            K.gather_rows(x, H, pt["pos0"], pt["sel0"], H, B, 1, L, H, row_off=ro)
            self._nt(pt["sel0"], self.P("bert.pooler.dense.weight"), pt["pooled"], B, H, H, bias=self.P("bert.pooler.dense.bias"), act=K.ACT_TANH)
            K.pretext_fwd(ws["vis_h"], ws["vispe_h"], pt["pooled"], st_vmp, pt["probs"], pt["sample"], pt["loss"], B, Nv, st_vmp.shape[1], H)
        return st

    def pretext_loss(self, st):
        """[1] f32 view holding the vis_pretext_loss of this forward (modeling.py:1131)."""
        return st.pretext[0]["loss"]

    def pooled_output(self, st):
        return st.pretext[0]["pooled"]

    def mlm_loss(self, st, labels, weights, drop_worst_ratio):
        """modeling.py:1083-1111 on the logits of this forward; returns a [1] f32 view holding the loss."""
        ws = st.ws
        V = self._model().config.vocab_size
        st.task_labels = labels.contiguous()
        K.mlm_loss_fwd(ws["logits"], ws["Vp"], st.task_labels, weights.to(torch.long).contiguous(), ws["loss"], ws["lse_ce"], ws["coef"],
                       ws["row_loss"], st.B, st.P, V, drop_worst_ratio=float(drop_worst_ratio))
        return ws["loss"][0:1]

    def vqa_loss(self, st, ans_labels):
        """modeling.py:1140: BCEWithLogits(mean) * num_answers."""
        ws = st.ws
        st.task_labels = ans_labels.to(torch.float32).contiguous()
        NA = self._model().num_answers
        K.bce_loss_fwd(ws["vq_logits"], ws["NAp"], st.task_labels, st.task_labels.stride(0), st.B, NA, ws["loss"])
        return ws["loss"][0:1]

    def mlm_logits(self, st):
        V = self._model().config.vocab_size
        return st.ws["logits"][:, :V].view(st.B, st.P, V)

    def vqa_logits(self, st):
        return st.ws["vq_logits"][:, :self._model().num_answers]

    def sequence_output(self, st):
        """[B, L, H] hidden states of the last layer.  After a packed forward the dropped (padding) positions hold zeros -- the dense
        run computes values there that nothing consumes."""
        cfg = self._model().config
        x = st.ws["layers"][cfg.num_hidden_layers - 1]["x2"]
        if st.pk is not None:
            dense = torch.zeros(st.B * st.L, cfg.hidden_size, device=x.device, dtype=x.dtype)
            K.rows_unpack(x, st.pk[1], st.pk[2], dense, cfg.hidden_size)
            x = dense
        return x.view(st.B, st.L, cfg.hidden_size)

    # ------------------------------------------------------------------------------------------
    # incremental greedy decoding with a K/V cache (modeling.py:1189-1253, :856-875, :386-394)
    # ------------------------------------------------------------------------------------------
    def _decode_workspace(self, B, T0, Lcap, Kb=1):
        key = ("dec", B, T0, Lcap, Kb)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        model = self._model()
        cfg = model.config
        H, I, NL, Nv, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, model.len_vis_input, cfg.vocab_size
        R = B * Kb                                    # sequences decoded in parallel after the first step
        M, Mv, dev = max(B * T0, R * 2), B * Nv, self.device

        def h(*s):
            return torch.empty(*s, device=dev, dtype=torch.float16)

        def f(*s):
            return torch.empty(*s, device=dev, dtype=torch.float32)

        def i64(*s):
            return torch.empty(*s, device=dev, dtype=torch.long)

        Vp = _ru(V, 64)
        ws = dict(Vp=Vp, maskb=torch.empty(max(B * T0, R * 2) * _ru(Lcap, 32), device=dev, dtype=torch.uint8),
                  img16=h(Mv, 2048), vpe_in=h(Mv, PE_PAD), wpe_pad=h(H, PE_PAD), h1=h(Mv, 2048), vis_h=h(Mv, H), vispe_h=h(Mv, H),
                  emb_pre=h(M, H), xa=h(M, H), xb=h(M, H), qkv=h(M, 3 * H), ctx=h(M, H), pre=h(M, H), x1=h(M, H), g=h(M, I),
                  kv=[h(B, Lcap, 2 * H) for _ in range(NL)],        # per layer: K | V of every position decoded so far
                  sel=h(R, H), tg=h(R, H), tln=h(R, H), logits=h(R, Vp), xids=i64(R, 2),
                  last_first=torch.full((R, 1), T0 - 1, device=dev, dtype=torch.long), last_step=torch.full((R, 1), 1, device=dev, dtype=torch.long),
                  # static copies of the caller's per-position inputs, so that the launches of a token step have call-invariant arguments
                  mask_static=i64(R, Lcap, Lcap), tt_steps=i64(Lcap, R, 2), pid_steps=i64(Lcap, R, 2),
                  out_ids=i64(B, Lcap), out_val=f(B, Lcap), plans={}, calls=0, plan_stream=None,
                  slab=f(self.DEC_SPLITS, R * 2, H),      # fp32 split-K partial sums of the token-step out-projection / FFN-down (vlp_dec_gemm -> vlp_dec_reduce_ln)
                  sk_ws=torch.empty(K.gemm_nt_splitk_workspace_bytes(min(M, 1024), max(I, 3 * H), max(self.SKINNY_SPLITS)), device=dev,
                                    dtype=torch.uint8))
        if Kb > 1:
            # beams: two caches per layer (select_beam_items permutes rows: gather from one into the other, then swap)
            ws.update(kvA=[h(R, Lcap, 2 * H) for _ in range(NL)], kvB=[h(R, Lcap, 2 * H) for _ in range(NL)],
                      kk_s=f(R, Kb), kk_i=i64(R, Kb), src_rows=i64(R), tot=f(Lcap, B, Kb), wids=i64(Lcap, B, Kb), ptrs=i64(Lcap, B, Kb),
                      eos=f(Lcap, B, Kb))
        self._ws[key] = ws
        return ws

    def _decode_check(self, vis_feats, vis_pe, input_ids, token_type_ids, attention_mask):
        model = self._model()
        cfg = model.config
        H, A, Nv = cfg.hidden_size, cfg.num_attention_heads, model.len_vis_input
        B, in_len = input_ids.shape
        out_len = token_type_ids.shape[1]
        if H != A * 64:
            raise RuntimeError("vlp_amd: attention kernels need head_dim == 64 (hidden %d, heads %d)" % (H, A))
        if vis_feats.shape[1] != Nv or vis_feats.shape[2] != 2048 or vis_pe.shape[2] != PE_DIM:
            raise RuntimeError("vlp_amd: expected vis_feats [B,%d,2048] and vis_pe [B,%d,%d]" % (Nv, Nv, PE_DIM))
        if in_len < Nv + 2 or out_len <= in_len or out_len > 256:
            raise RuntimeError("vlp_amd: decode needs %d <= input length < output length <= 256 (got %d, %d)" % (Nv + 2, in_len, out_len))
        if attention_mask.dim() != 3 or attention_mask.shape[1] < out_len or attention_mask.shape[2] < out_len:
            raise RuntimeError("vlp_amd: decode expects a [B, L, L] attention mask covering the output length")
        return B, in_len, out_len

    def _decode_regions(self, ws, vis_feats, vis_pe):
        """Region projections, once per decode call (:1192-1193)."""
        model = self._model()
        H, Nv = model.config.hidden_size, model.len_vis_input
        Mv = vis_feats.shape[0] * Nv
        vf = vis_feats.reshape(Mv, 2048)
        if vf.dtype == torch.float32:
            K.copy2d(vf.contiguous(), 2048, True, ws["img16"], 2048, Mv, 2048, 2048)
            img = ws["img16"]
        else:
            img = vf.contiguous()
        vp = vis_pe.reshape(Mv, PE_DIM).contiguous()
        K.copy2d(vp, PE_DIM, vp.dtype == torch.float32, ws["vpe_in"], PE_PAD, Mv, PE_DIM, PE_PAD)
        K.copy2d(self.P("vis_pe_embed.0.weight"), PE_DIM, False, ws["wpe_pad"], PE_PAD, H, PE_DIM, PE_PAD)
        self._nt(img, self.P("vis_embed.0.weight"), ws["h1"], Mv, 2048, 2048, bias=self.P("vis_embed.0.bias"), act=K.ACT_RELU)
        self._nt(ws["h1"], self.P("vis_embed.2.weight"), ws["vis_h"], Mv, H, 2048, bias=self.P("vis_embed.2.bias"), act=K.ACT_RELU)
        self._nt(ws["vpe_in"], ws["wpe_pad"], ws["vispe_h"], Mv, H, PE_PAD, bias=self.P("vis_pe_embed.0.bias"), act=K.ACT_RELU)

    # Token steps on the burst kernels of csrc/decode.hip (round 6): 7 launches per layer instead of 12 (vlp_dec_gemm QKV with the K/V append
    # fused, attention, out-projection as 4 k slices, slab reduce + bias + residual + LayerNorm in one launch, FFN-up + GeLU, FFN-down as 4 k
    # slices, reduce + LayerNorm).  VLP_DECODE_FUSED=0: the round-2 path (split-K skinny GEMMs, separate reduces / kv_append / LayerNorms).
    DECODE_FUSED = os.environ.get("VLP_DECODE_FUSED", "1") == "1"
    DEC_SPLITS = 4
    # VLP_DECODE_LN_PROLOGUE=1: the attention-output LayerNorm as a PROLOGUE of the FFN-up launch (6 launches per layer).  Built, tested and measured in
    # round 6, NOT the default: every one of the 192 workgroups of the FFN-up grid normalises its 64 rows itself (16 rows per wave, ~25 VALU ops per
    # element on one wave per SIMD): the launch goes 6.4 -> 15.5 us, more than the 4.8 us LayerNorm launch + boundary it removes
    # (0.645 vs 0.559 ms per token step, profiles/r06_decode_ln_prologue_ab.txt).
    DEC_LN_PROLOGUE = os.environ.get("VLP_DECODE_LN_PROLOGUE", "0") == "1"
    DEC_VOCAB = os.environ.get("VLP_DECODE_VOCAB_BURST", "1") == "1"      # the tied vocabulary projection of a token step on vlp_dec_gemm too

    def _decode_layers_fused(self, ws, caches, Lcap, x, alt, maskb, R, T, st, prefix):
        """The 12 BertLayers of a token step (M = R * T <= a few hundred rows) on vlp_dec_gemm / vlp_dec_reduce_ln.  Same arithmetic and
        rounding points as the unfused path (fp16 GEMM outputs, fp16 pre-LayerNorm sums, fp32 LayerNorm statistics); only the fp32
        summation order of the contractions differs."""
        model = self._model()
        cfg = model.config
        H, I, A, NL = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_hidden_layers
        scale = 1.0 / math.sqrt(H // A)
        Lk, M, S = st + T, R * T, self.DEC_SPLITS
        slab = ws["slab"].view(-1)[:S * M * H].view(S, M, H)        # slab s holds rows [s * M, (s + 1) * M)
        for i in range(NL):
            Ln = "bert.encoder.layer.%d." % i
            kv = caches[i]
            K.dec_gemm(x, self.P(Ln + "attention.self.query.weight"), M, 3 * H, H, y=ws["qkv"], bias=self.P(Ln + "attention.self.query.bias"),
                       kv_cache=kv, kv_col0=H, kv_Lcap=Lcap, kv_T=T, kv_start=st)
            if prefix is None:
                K.attn_decode(ws["qkv"], 3 * H, T, kv, kv[:, :, H:], 2 * H, Lcap, maskb, ws["ctx"], R, T, Lk, A, scale)
            else:
                pk = prefix[0][i]
                K.attn_decode(ws["qkv"], 3 * H, T, kv, kv[:, :, H:], 2 * H, Lcap, maskb, ws["ctx"], R, T, Lk, A, scale, k_prefix=pk,
                              v_prefix=pk[:, :, H:], prefix_rows=Lcap, n_prefix=prefix[1], beams=prefix[2])
            if self.DEC_LN_PROLOGUE:
                # out-projection + bias + residual -> fp16 pre-LayerNorm rows; FFN-up normalises them in its prologue (and writes x1 for the next residual)
                K.dec_gemm(ws["ctx"], self.P(Ln + "attention.output.dense.weight"), M, H, H, y=ws["pre"], bias=self.P(Ln + "attention.output.dense.bias"), residual=x)
                K.dec_gemm(ws["pre"], self.P(Ln + "intermediate.dense.weight"), M, I, H, y=ws["g"], bias=self.P(Ln + "intermediate.dense.bias"), act=K.ACT_GELU,
                           ln_gamma=self.P(Ln + "attention.output.LayerNorm.weight"), ln_beta=self.P(Ln + "attention.output.LayerNorm.bias"), ln_out=ws["x1"])
            else:
                K.dec_gemm(ws["ctx"], self.P(Ln + "attention.output.dense.weight"), M, H, H, slab=slab, splits=S)
                K.dec_reduce_ln(slab, S, self.P(Ln + "attention.output.dense.bias"), x, self.P(Ln + "attention.output.LayerNorm.weight"),
                                self.P(Ln + "attention.output.LayerNorm.bias"), ws["x1"], M, H)
                K.dec_gemm(ws["x1"], self.P(Ln + "intermediate.dense.weight"), M, I, H, y=ws["g"], bias=self.P(Ln + "intermediate.dense.bias"), act=K.ACT_GELU)
            K.dec_gemm(ws["g"], self.P(Ln + "output.dense.weight"), M, H, I, slab=slab, splits=S)
            K.dec_reduce_ln(slab, S, self.P(Ln + "output.dense.bias"), ws["x1"], self.P(Ln + "output.LayerNorm.weight"),
                            self.P(Ln + "output.LayerNorm.bias"), alt, M, H)
            x, alt = alt, x
        return x

    def _decode_model_step(self, ws, caches, Lcap, xids, tt, pid, mask_view, R, T, st, first, prefix=None):
        """One incremental forward of R sequences x T new tokens at absolute positions st..st+T-1: Q/K/V projection of the new
        tokens, K|V appended to the per-layer caches, attention over positions 0..st+T-1, LM head on the last ([MASK]) slot.
        Leaves the logits [R, V] in ws['logits'] (pitch ws['Vp']).  prefix = (per-sample caches, n_prefix, beams): beam search keeps
        the positions < n_prefix (regions + [SEP], identical for all beams of a sample) once per sample; `caches` then hold only
        the generated positions of every beam."""
        model = self._model()
        cfg = model.config
        H, I, A, NL, Nv, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_hidden_layers, model.len_vis_input, cfg.vocab_size
        E, C = "bert.embeddings.", "cls.predictions."
        scale = 1.0 / math.sqrt(H // A)
        Lk = st + T
        Lkp = _ru(Lk, 32)
        M = R * T
        maskb = ws["maskb"][:R * T * Lkp]
        K.mask_pack_rect(mask_view, maskb, R, T, Lk, Lkp)
        K.embed_fwd(xids, tt, self.P(E + "word_embeddings.weight"), self.P(E + "position_embeddings.weight"),
                    self.P(E + "token_type_embeddings.weight"), ws["vis_h"], ws["vispe_h"], ws["emb_pre"], R, T, Nv if first else 0, H,
                    position_ids=pid)
        x, alt = ws["xa"], ws["xb"]
        K.layernorm_fwd(ws["emb_pre"], self.P(E + "LayerNorm.weight"), self.P(E + "LayerNorm.bias"), x, M, H)
        def same_in_all_layers(suffix):
            return lambda: [self.P("bert.encoder.layer.%d.%s" % (j, suffix)) for j in range(NL)]
        fused = (self.DECODE_FUSED and not first and M <= 1024 and H % 256 == 0 and H <= 768 and I % (64 * self.DEC_SPLITS) == 0 and
                 I // self.DEC_SPLITS <= 768 and H % (64 * self.DEC_SPLITS) == 0)
        if fused:
            x = self._decode_layers_fused(ws, caches, Lcap, x, alt, maskb, R, T, st, prefix)
        for i in range(0 if not fused else NL, NL):
            Ln = "bert.encoder.layer.%d." % i
            kv = caches[i]
            self._nt_skinny(x, self.P(Ln + "attention.self.query.weight"), ws["qkv"], M, 3 * H, H, ws["sk_ws"], tune_ws=same_in_all_layers("attention.self.query.weight"), bias=self.P(Ln + "attention.self.query.bias"))
            K.kv_append(ws["qkv"], 3 * H, kv, Lcap, R, T, st, H)
            if prefix is None:
                K.attn_decode(ws["qkv"], 3 * H, T, kv, kv[:, :, H:], 2 * H, Lcap, maskb, ws["ctx"], R, T, Lk, A, scale)
            else:
                pk = prefix[0][i]
                K.attn_decode(ws["qkv"], 3 * H, T, kv, kv[:, :, H:], 2 * H, Lcap, maskb, ws["ctx"], R, T, Lk, A, scale, k_prefix=pk,
                              v_prefix=pk[:, :, H:], prefix_rows=Lcap, n_prefix=prefix[1], beams=prefix[2])
            self._nt_skinny(ws["ctx"], self.P(Ln + "attention.output.dense.weight"), ws["pre"], M, H, H, ws["sk_ws"], tune_ws=same_in_all_layers("attention.output.dense.weight"), bias=self.P(Ln + "attention.output.dense.bias"),
                     residual=x)
            K.layernorm_fwd(ws["pre"], self.P(Ln + "attention.output.LayerNorm.weight"), self.P(Ln + "attention.output.LayerNorm.bias"),
                            ws["x1"], M, H)
            self._nt_skinny(ws["x1"], self.P(Ln + "intermediate.dense.weight"), ws["g"], M, I, H, ws["sk_ws"], tune_ws=same_in_all_layers("intermediate.dense.weight"), bias=self.P(Ln + "intermediate.dense.bias"),
                     act=K.ACT_GELU)
            self._nt_skinny(ws["g"], self.P(Ln + "output.dense.weight"), ws["pre"], M, H, I, ws["sk_ws"], tune_ws=same_in_all_layers("output.dense.weight"), bias=self.P(Ln + "output.dense.bias"), residual=ws["x1"])
            K.layernorm_fwd(ws["pre"], self.P(Ln + "output.LayerNorm.weight"), self.P(Ln + "output.LayerNorm.bias"), alt, M, H)
            x, alt = alt, x
        # ---- LM head on the [MASK] slot (:1226-1228 / :1293-1296) ----------------------------------
        if fused:
            # the [MASK] slot is the LAST of the T new rows of every sequence: a strided view of x (row pitch T * H), no gather launch
            sel = x.view(-1)[:R * T * H].view(R, T * H)[:, (T - 1) * H:]
            K.dec_gemm(sel, self.P(C + "transform.dense.weight"), R, H, H, y=ws["tg"], bias=self.P(C + "transform.dense.bias"), act=K.ACT_GELU)
        else:
            K.gather_rows(x, H, ws["last_first"] if first else ws["last_step"], ws["sel"], H, R, 1, T, H)
            self._nt(ws["sel"], self.P(C + "transform.dense.weight"), ws["tg"], R, H, H, bias=self.P(C + "transform.dense.bias"), act=K.ACT_GELU)
        K.layernorm_fwd(ws["tg"], self.P(C + "transform.LayerNorm.weight"), self.P(C + "transform.LayerNorm.bias"), ws["tln"], R, H)
        if fused and self.DEC_VOCAB:
            K.dec_gemm(ws["tln"], self.P("bert.embeddings.word_embeddings.weight"), R, V, H, y=ws["logits"], bias=self.P(C + "bias"))
        else:
            self._nt(ws["tln"], self.P("bert.embeddings.word_embeddings.weight"), ws["logits"], R, V, H, bias=self.P(C + "bias"), ldy=ws["Vp"])

    def _decode_static_inputs(self, ws, token_type_ids, position_ids, attention_mask, in_len, out_len, rep):
        """Copies the caller's per-position inputs of the token steps s >= 1 into workspace buffers (one torch op each), so that the
        launches of those steps take the same arguments on every call (launch plans).  rep = beams per sample (first_expand)."""
        n = out_len - in_len - 1                     # steps 1 .. n_steps-1 use the position window (in_len+s-1, in_len+s)
        if n > 0:
            tt = token_type_ids.unfold(1, 2, 1)[:, in_len:in_len + n].transpose(0, 1)      # [n, B, 2]
            pid = position_ids.unfold(1, 2, 1)[:, in_len:in_len + n].transpose(0, 1)
            if rep > 1:
                tt, pid = tt.repeat_interleave(rep, dim=1), pid.repeat_interleave(rep, dim=1)
            ws["tt_steps"][1:n + 1].copy_(tt)
            ws["pid_steps"][1:n + 1].copy_(pid)
        am = attention_mask[:, :out_len, :out_len]
        ws["mask_static"].copy_(am.repeat_interleave(rep, dim=0) if rep > 1 else am)

    DECODE_GRAPHS = os.environ.get("VLP_DECODE_GRAPHS", "1") == "1"

    def _run_planned(self, ws, key, fn):
        """Token steps s >= 1 launch ~200 small kernels with call-invariant arguments; at ~4.5 us of HIP launch cost each the host, not
        the GPU, bounded decoding.  Call 1 of a workspace runs (and autotunes) normally; call 2 captures each step into a hipGraph
        (torch.cuda.CUDAGraph over the launches our C ABI puts on the capture stream) and replays it; later calls only replay.
        If capture is unavailable the recorded ctypes plan (no Python argument marshalling) is the fallback."""
        stream = torch.cuda.current_stream().cuda_stream
        item = ws["plans"].get(key)
        if item is not None and ws["plan_stream"] == stream:
            if isinstance(item, list):
                K.replay(item)
            else:
                item.replay()
            return
        if ws["calls"] == 0:
            fn()
            return
        ws["plan_stream"] = stream
        if self.DECODE_GRAPHS:
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    fn()
                g.replay()
                ws["plans"][key] = g
                return
            except Exception as e:               # capture not possible here: fall back to the ctypes plan (host-bound: say so, once)
                torch.cuda.synchronize()
                Engine.DECODE_GRAPHS = False
                import warnings
                warnings.warn("vlp_amd: hipGraph capture of a decoder token step failed (%r); token steps fall back to replayed ctypes launch plans, "
                              "which are bound by host launch overhead" % (e,), VlpPerformanceWarning)
        with K.record() as plan:
            fn()
        ws["plans"][key] = plan

    def decode_greedy(self, vis_feats, vis_pe, input_ids, token_type_ids, position_ids, attention_mask, mask_word_id, sample=False):
        """Greedy incremental decoding.  Step s feeds the tokens that are new since step s-1 plus one [MASK] slot, projects them
        to Q/K/V, appends K|V to the per-layer cache at their absolute positions (the [MASK] slot's entry is overwritten by the
        real token in the next step, which is exactly what the reference's hidden-state history achieves by dropping the last
        row, :1236-1247) and attends over the cache.  Returns (ids [B, n] int64, max logits [B, n] f32); with sample=True the
        ids are drawn from softmax(logits) (:1229-1235) and the second output holds their log-probabilities."""
        self.pack()
        self.wait_params()
        V = self._model().config.vocab_size
        B, in_len, out_len = self._decode_check(vis_feats, vis_pe, input_ids, token_type_ids, attention_mask)
        n_steps, T0 = out_len - in_len, in_len + 1
        ws = self._decode_workspace(B, T0, out_len)
        dev = self.device
        attention_mask = attention_mask.to(torch.long)
        token_type_ids, position_ids = token_type_ids.to(torch.long), position_ids.to(torch.long)
        self._decode_regions(ws, vis_feats, vis_pe)
        self._decode_static_inputs(ws, token_type_ids, position_ids, attention_mask, in_len, out_len, 1)
        out_ids, out_val, am = ws["out_ids"], ws["out_val"], ws["mask_static"]
        x_first = torch.cat((input_ids.to(torch.long), torch.full((B, 1), int(mask_word_id), device=dev, dtype=torch.long)), dim=1).contiguous()
        ws["xids"][:, 1] = int(mask_word_id)

        def pick(s):
            if sample:
                self.step_seed += 1
                for dst in (out_ids[:, s], ws["xids"][:, 0]):                   # same (seed, stream) -> same draw; 2nd = next input token
                    K.sample_rows(ws["logits"], ws["Vp"], B, V, self.base_seed + self.step_seed, 7001, dst, out_val[:, s])
            else:
                K.argmax_rows2(ws["logits"], ws["Vp"], B, V, out_ids[:, s], ws["xids"][:, 0], out_val[:, s])      # + next step's first input token

        # step 0: the whole prefix + [MASK]
        self._decode_model_step(ws, ws["kv"], out_len, x_first, token_type_ids[:, :T0].contiguous(), position_ids[:, :T0].contiguous(),
                                am[:, :T0, :T0], B, T0, 0, True)
        pick(0)
        for s in range(1, n_steps):
            st = in_len + s - 1

            def step(s=s, st=st):
                self._decode_model_step(ws, ws["kv"], out_len, ws["xids"], ws["tt_steps"][s], ws["pid_steps"][s], am[:, st:st + 2, :st + 2],
                                        B, 2, st, False)
                pick(s)
            if sample:
                step()                       # the seed changes every step: not plannable
            else:
                self._run_planned(ws, ("greedy", s), step)
        ws["calls"] += 1
        return out_ids[:, :n_steps].clone(), out_val[:, :n_steps].clone()

    def decode_beam(self, vis_feats, vis_pe, input_ids, token_type_ids, position_ids, attention_mask, mask_word_id, beam_size, eos_id,
                    min_len=0, forbid_fn=None):
        """Beam search frames (modeling.py:1255-1430) on the K/V-cache decoder.  The first step runs B sequences; its cache rows are
        replicated to the B*K beams (first_expand), afterwards every step decodes B*K sequences and the caches follow the back
        pointers (select_beam_items) -- only the generated positions; the prefix is identical for all beams of a sample and is kept,
        and streamed by the attention kernel, once per sample.
        `forbid_fn(step_ids [B,K] list, back_ptrs [B,K] list, first)` -> uint8 [B*K, V] numpy mask or None implements the host-side
        n-gram blocking (:1367-1430); when given, ids / pointers are copied to the host every step exactly as the reference does.
        Returns (total_scores, step_ids, back_ptrs) as [frames, B, K] tensors on the device."""
        self.pack()
        self.wait_params()
        model = self._model()
        cfg = model.config
        H, NL, V = cfg.hidden_size, cfg.num_hidden_layers, cfg.vocab_size
        Kb = int(beam_size)
        B, in_len, out_len = self._decode_check(vis_feats, vis_pe, input_ids, token_type_ids, attention_mask)
        if Kb < 2 or Kb > 64:
            raise RuntimeError("vlp_amd: beam size must be in [2, 64]")
        n_steps, T0 = out_len - in_len, in_len + 1
        R = B * Kb
        ws = self._decode_workspace(B, T0, out_len, Kb)
        dev = self.device
        attention_mask = attention_mask.to(torch.long)
        if attention_mask.stride(2) != 1:
            attention_mask = attention_mask.contiguous()
        token_type_ids, position_ids = token_type_ids.to(torch.long), position_ids.to(torch.long)
        self._decode_regions(ws, vis_feats, vis_pe)
        self._decode_static_inputs(ws, token_type_ids, position_ids, attention_mask, in_len, out_len, Kb)     # first_expand (:1361-1365)
        tot, wids, ptrs, eos, am = ws["tot"], ws["wids"], ws["ptrs"], ws["eos"], ws["mask_static"]
        x_first = torch.cat((input_ids.to(torch.long), torch.full((B, 1), int(mask_word_id), device=dev, dtype=torch.long)), dim=1).contiguous()
        ws["xids"][:, 1] = int(mask_word_id)
        bufs = (ws["kvA"], ws["kvB"])
        forbid = None

        def frame(s, rows, first, forbid, block_eos):
            K.logsoftmax_topk(ws["logits"], ws["Vp"], rows, V, Kb, ws["kk_s"], ws["kk_i"], forbid=forbid, eos_id=int(eos_id), block_eos=block_eos)
            K.beam_select(ws["kk_s"], ws["kk_i"], None if first else tot[s - 1], None if first else eos[s - 1], tot[s], wids[s], ptrs[s], eos[s],
                          ws["src_rows"], ws["xids"][:, 0], B, Kb, first, int(eos_id))

        # step 0: B sequences.  first_expand (:1325-1332) is implicit: the prefix K|V of a sample stays in its per-sample cache and is
        # shared by its beams inside the attention kernel; the per-beam caches only ever hold generated positions
        self._decode_model_step(ws, ws["kv"], out_len, x_first, token_type_ids[:, :T0].contiguous(), position_ids[:, :T0].contiguous(),
                                attention_mask[:, :T0, :T0], B, T0, 0, True)
        frame(0, B, True, None, bool(min_len) and (1 <= min_len))
        prefix = (ws["kv"], in_len, Kb)
        if forbid_fn is not None:
            fm = forbid_fn(wids[0].tolist(), ptrs[0].tolist(), True)
            forbid = None if fm is None else torch.from_numpy(fm).to(dev)
        for s in range(1, n_steps):
            st = in_len + s - 1
            cur, other = bufs[(s - 1) & 1], bufs[s & 1]            # the caches swap after every step >= 1
            block_eos = bool(min_len) and (s + 1 <= min_len)

            def step(s=s, st=st, cur=cur, other=other, block_eos=block_eos, forbid=forbid):
                self._decode_model_step(ws, cur, out_len, ws["xids"], ws["tt_steps"][s], ws["pid_steps"][s], am[:, st:st + 2, :st + 2], R, 2, st,
                                        False, prefix=prefix)
                frame(s, R, False, forbid, block_eos)
                if s + 1 < n_steps:      # select_beam_items (:1334-1359): generated positions in_len .. st follow their beam
                    for i in range(NL):
                        K.kv_gather(cur[i], out_len, other[i], out_len, ws["src_rows"], R, in_len, st + 1, 2 * H)
            if forbid_fn is not None:
                step()                   # the forbid mask is a fresh tensor every step: not plannable
                fm = forbid_fn(wids[s].tolist(), ptrs[s].tolist(), False)
                forbid = None if fm is None else torch.from_numpy(fm).to(dev)
            else:
                self._run_planned(ws, ("beam", s, int(eos_id), block_eos), step)
        ws["calls"] += 1
        return tot[:n_steps].clone(), wids[:n_steps].clone(), ptrs[:n_steps].clone()

    # ------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------
    def _tn_splits(self, a, b, c, M, N, Kd, ws):
        if self.TN_SPLITS is not None:
            return (self.GEMM_TN_VARIANT, self.TN_SPLITS)
        key = (M, N, Kd)
        sp = Engine._tn_choice.get(key)
        if sp is not None:
            return sp
        if not tuning.AUTOTUNE:
            sp = Engine._tn_choice[key] = tuning.tn_choice(M, N, Kd)
            return sp
        best, best_t = (self.GEMM_TN_VARIANT, 0), float("inf")
        if M >= 1024:
            torch.cuda.synchronize()
            scratch = torch.empty(N, Kd, device=c.device, dtype=torch.float16)   # never time into the live gradient buffer
            for rnd in range(2):
                for var in self.TN_VARIANT_CANDIDATES:
                    for cand in self.TN_SPLIT_CANDIDATES:
                        if cand > 1 and M // cand < 128:
                            continue
                        K.gemm_tn(a, b, scratch, M, N, Kd, beta=0, workspace=ws["tn_ws"], variant=var, splits=cand)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(3):
                            K.gemm_tn(a, b, scratch, M, N, Kd, beta=0, workspace=ws["tn_ws"], variant=var, splits=cand)
                        e1.record()
                        e1.synchronize()
                        t = e0.elapsed_time(e1)
                        if t < best_t:
                            best, best_t = (var, cand), t
        Engine._tn_choice[key] = best
        tuning.remember("tn", M, N, Kd, best)
        return best

    def _tn(self, a, b, c, M, N, Kd, ws, beta, bias=None, ws_key="tn_ws", **kw):
        """wgrad GEMM; `bias` (the Linear's bias gradient = column sums of dY) is fused into the same launch.  ws_key: the split-M scratch
        (launches on the side stream share "tn_ws"; the one TN launch of the main stream's head dgrad has its own)."""
        var, sp = self._tn_splits(a, b, c, M, N, Kd, ws)
        K.gemm_tn(a, b, c, M, N, Kd, beta=beta, workspace=ws[ws_key], variant=var, bias_out=bias, splits=sp, **kw)

    def _bucket_done(self, idx):
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(idx)

    def backward(self, st, gscale, task, g_pretext=None):
        """gscale: device f32 tensor [1] = upstream gradient of the task loss (x loss scale); g_pretext: the same for the pretext loss of
        a mask_image_regions forward.  Writes every parameter gradient into the flat gradient buffers."""
        if st.gen != self.gen:
            raise RuntimeError("vlp_amd: the activations of this forward were overwritten by a later forward "
                               "(one in-flight forward per model; run backward before the next forward)")
        model = self._model()
        cfg = model.config
        H, I, A, NL, Nv, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_hidden_layers, model.len_vis_input, cfg.vocab_size
        B, L, P, ws, seed = st.B, st.L, st.P, st.ws, st.seed
        p, pa = st.p_drop
        M, Mv = B * L, B * Nv
        ro = rm = None
        if st.pk is not None:
            ro, rm, M = st.pk                # packed rows: every [M, *] activation of this step has M' = sum of the kept lengths rows
        beta = 1 if self.grads_dirty else 0
        if beta and self.shard_plan is not None:
            raise RuntimeError("vlp_amd: VLP_DDP_MODE=sharded keeps only this rank's chunk of the reduced gradient, so gradients cannot be "
                               "accumulated over several backward passes (--gradient_accumulation_steps > 1); use allreduce or rs_ag")
        img, input_ids, token_type_ids, masked_pos = st.batch
        self.wait_params()           # the optimizer stream has read the previous gradients and written every parameter
        if getattr(self, "_shadow_ev", None) is not None:
            torch.cuda.current_stream().wait_event(self._shadow_ev)     # transposed during the forward (side stream)
            self._shadow_ev = None
        else:
            self._refresh_shadows()
        sh = self._shadows()
        x_last = ws["layers"][NL - 1]["x2"]
        dx = ws["dx"]
        dx.zero_()
        E = "bert.embeddings."
        if beta == 0:
            # tables that are only ever accumulated into (+=) start from zero
            self.G(E + "position_embeddings.weight").zero_()
            self.G(E + "token_type_embeddings.weight").zero_()

        main = torch.cuda.current_stream()
        use_side = self.WGRAD_SIDE_STREAM
        if use_side and self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._side_done = [None, None]
        if use_side and getattr(self, "_side_done", None) is None:
            self._side_done = [None, None]
        side = self._side if use_side else None

        def on_side(fn):
            if not use_side:
                fn()
                return
            side.wait_stream(main)
            with torch.cuda.stream(side):
                fn()

        if use_side:
            self._side_busy = True
        head_wgrads = []        # weight gradients of the task head: issued on the side stream once the head's dgrad chain is on the main stream

        # ---- heads ------------------------------------------------------------------------------------
        if task == "vqa2":
            NA, NAp = model.num_answers, ws["NAp"]
            K.bce_loss_bwd(ws["vq_logits"], NAp, st_labels(st), st_labels(st).stride(0), B, NA, gscale, ws["vq_dlogits"], NAp)
            self._tn(ws["vq_dlogits"], ws["vq_a1"], self.G("ans_classifier.2.weight"), B, NA, 2 * H, ws, beta, bias=self.G("ans_classifier.2.bias"))
            self._nt(ws["vq_dlogits"], sh["a2T"], ws["vq_dz1"], B, 2 * H, NAp, mul_src=ws["vq_a1"], mul_mode=K.MUL_RELU_MASK)
            self._tn(ws["vq_dz1"], ws["vq_e"], self.G("ans_classifier.0.weight"), B, 2 * H, H, ws, beta, bias=self.G("ans_classifier.0.bias"))
            self._nt(ws["vq_dz1"], sh["a0T"], ws["vq_de"], B, H, 2 * H)
            K.vqa_mul_bwd(x_last, ws["vq_de"], dx, B, L, Nv, H, row_off=ro)
            if beta == 0:
                self.G(E + "word_embeddings.weight").zero_()     # no tied-decoder wgrad in this task: scatter needs zeros
        elif not st.has_mlm:
            if beta == 0:                                        # empty masked_pos (modeling.py:1096-1098): only the pretext loss is live
                self.G(E + "word_embeddings.weight").zero_()
                for n in ("bias", "transform.dense.weight", "transform.dense.bias", "transform.LayerNorm.weight", "transform.LayerNorm.bias"):
                    self.G("cls.predictions." + n).zero_()
        else:
            C = "cls.predictions."
            R, Vp = B * P, ws["Vp"]
            K.mlm_loss_bwd(ws["logits"], Vp, st_labels(st), ws["lse_ce"], ws["coef"], gscale, ws["dlogits"], Vp, R, V)
            # tied decoder (modeling.py:445-448): dE[V,H] = dlogits^T . t ; the embedding scatter adds to it later
            head_wgrads.append(lambda: self._tn(ws["dlogits"], ws["tln"], self.G(E + "word_embeddings.weight"), R, V, H, ws, beta, bias=self.G(C + "bias")))
            # dgrad through the tied decoder: dt[R,H] = dlogits[R,V] . E[V,H].  As an NT GEMM this is 12 workgroups walking
            # K = 29056; instead transpose dlogits (11 MB) and contract over the vocabulary rows with the split-M wgrad kernel,
            # reading E in place (no E^T shadow).
            Rp = _ru(R, 64)
            K.transpose(ws["dlogits"], Vp, ws["dlT"], Rp, R, Vp, Rp)
            self._tn(ws["dlT"], self.P(E + "word_embeddings.weight"), ws["dtln"], V, R, H, ws, 0, ws_key="tn_ws_main")
            K.layernorm_bwd(ws["dtln"], ws["tg"], self.P(C + "transform.LayerNorm.weight"), ws["tstat"][0], ws["tstat"][1], ws["dtg"],
                            self.G(C + "transform.LayerNorm.weight"), self.G(C + "transform.LayerNorm.bias"), R, H, ws["ln_ws"], beta=beta)
            K.gelu_bwd(ws["dtg"], ws["tz"], ws["dtz"], R * H)
            head_wgrads.append(lambda: self._tn(ws["dtz"], ws["sel"], self.G(C + "transform.dense.weight"), R, H, H, ws, beta, bias=self.G(C + "transform.dense.bias")))
            self._nt(ws["dtz"], sh["tT"], ws["dsel"], R, H, H)
            K.scatter_add_rows(ws["dsel"], H, masked_pos.contiguous(), dx, H, B, P, L, H, row_off=ro)
        pt = st.pretext
        if pt is not None:
            # vis_pretext_loss backward (modeling.py:1113-1131): masked rows of d_vis_h / d_vispe_h (embed_bwd below leaves them alone),
            # then the pooler: tanh' is applied by the kernel, dW = d^T . h[:, 0], dh[:, 0] += d . W
            ptw, vmp = pt
            if g_pretext is None:
                g_pretext = torch.zeros(1, device=self.device, dtype=torch.float32)
            K.pretext_bwd(ws["vis_h"], ws["vispe_h"], ptw["pooled"], vmp, ptw["probs"], g_pretext, ws["d_vis_h"], ws["d_vispe_h"], ptw["dpool"],
                          B, Nv, vmp.shape[1], H, drop_p=p, seed=seed, vis_stream=1001, vispe_stream=1002)
            self._tn(ptw["dpool"], ptw["sel0"], self.G("bert.pooler.dense.weight"), B, H, H, ws, beta, bias=self.G("bert.pooler.dense.bias"))
            K.transpose(self.P("bert.pooler.dense.weight"), H, ptw["pT"], H, H, H, H)
            self._nt(ptw["dpool"], ptw["pT"], ptw["dsel0"], B, H, H)
            K.scatter_add_rows(ptw["dsel0"], H, ptw["pos0"], dx, H, B, 1, L, H, row_off=ro)
        elif beta == 0 and getattr(self, "_pooler_dirty", False):
            self.G("bert.pooler.dense.weight").zero_()           # a previous pretext step left gradients there; unused now
            self.G("bert.pooler.dense.bias").zero_()
        self._pooler_dirty = pt is not None

        def head_tail():
            for fn in head_wgrads:
                fn()
            self._bucket_done(0)
        if self.TAIL_ON_SIDE:
            on_side(head_tail)      # (the side stream is ordered behind everything the main stream has issued so far)
        else:
            head_tail()

        # ---- encoder layers, last to first ----------------------------------------------------------------
        # The dgrad chain (LN-bwd -> dgrad GEMMs -> attention-bwd) is the critical path; the four weight-gradient GEMMs of a
        # layer only consume its dY tensors.  They are issued on a side stream (ordered after their producers by
        # wait_stream) so that they co-run with the next kernels of the chain; dY buffers alternate by layer parity and the
        # main stream waits for the side stream's event before it reuses a set.
        dctx = ws["dctx"]
        scale = 1.0 / math.sqrt(H // A)
        slot_bytes = ws["ln_slot_bytes"]
        defer = self.LN_DEFER

        def ln_slot(k):
            return ws["ln_slots"][k * slot_bytes:(k + 1) * slot_bytes]

        grouped = self.GROUPED_WGRAD and M >= 2048       # enough rows for a long contraction per workgroup; tiny batches keep split-M
        for i in reversed(range(NL)):
            Ln = "bert.encoder.layer.%d." % i
            a = ws["layers"][i]
            s = sh["layers"][i]
            x_in = ws["layers"][i - 1]["x2"] if i > 0 else ws["x0"]
            ds = ws["dyset"][i & 1]
            if use_side and self._side_done[i & 1] is not None:
                main.wait_event(self._side_done[i & 1])       # wgrads of layer i+2 have finished reading this set
            # BertOutput: LN(dropout(dense(g)) + x1)   (modeling.py:353-357)
            dpre = ds["dpre2"]
            K.layernorm_bwd(dx, a["pre2"], self.P(Ln + "output.LayerNorm.weight"), a["st2"][0], a["st2"][1], dpre,
                            self.G(Ln + "output.LayerNorm.weight"), self.G(Ln + "output.LayerNorm.bias"), M, H, ln_slot(2 * i + 1), beta=beta,
                            dx_drop=ds["dpre2_d"] if p > 0 else None, out_drop=(p, seed, 16 * i + 3), defer_reduce=defer, row_map=rm)
            dy2 = ds["dpre2_d"] if p > 0 else dpre
            if not grouped:
                on_side(lambda: self._tn(dy2, a["g"], self.G(Ln + "output.dense.weight"), M, H, I, ws, beta, bias=self.G(Ln + "output.dense.bias")))
            self._nt(dy2, s["w2T"], ds["dz"], M, I, H, mul_src=a["z"], mul_mode=K.MUL_PLAIN)      # dG * gelu'(z) (stored by the forward)
            # BertIntermediate (modeling.py:340-343)
            if not grouped:
                on_side(lambda: self._tn(ds["dz"], a["x1"], self.G(Ln + "intermediate.dense.weight"), M, I, H, ws, beta,
                                         bias=self.G(Ln + "intermediate.dense.bias")))
            self._nt(ds["dz"], s["w1T"], dx, M, H, I, residual=dpre)                            # + residual path of LN2's input
            # BertSelfOutput: LN(dropout(dense(ctx)) + x)   (modeling.py:313-317)
            dpre = ds["dpre1"]
            K.layernorm_bwd(dx, a["pre1"], self.P(Ln + "attention.output.LayerNorm.weight"), a["st1"][0], a["st1"][1], dpre,
                            self.G(Ln + "attention.output.LayerNorm.weight"), self.G(Ln + "attention.output.LayerNorm.bias"), M, H, ln_slot(2 * i),
                            beta=beta, dx_drop=ds["dpre1_d"] if p > 0 else None, out_drop=(p, seed, 16 * i + 2), defer_reduce=defer, row_map=rm)
            dy1 = ds["dpre1_d"] if p > 0 else dpre
            if not grouped:
                on_side(lambda: self._tn(dy1, a["ctx"], self.G(Ln + "attention.output.dense.weight"), M, H, H, ws, beta,
                                         bias=self.G(Ln + "attention.output.dense.bias")))
            self._nt(dy1, s["oT"], dctx, M, H, H)
            # BertSelfAttention (modeling.py:268-303)
            dqkv = ds["dqkv"]
            K.attn_bwd(a["qkv"], ws["maskb"], ws["maskt"], a["ctx"], dctx, a["lse"], dqkv, ws["delta"], B, L, A, scale, dropout_p=pa, seed=seed,
                       rng_stream=16 * i + 1, row_off=ro)

            def last_wgrad():
                if grouped:
                    # the layer's four weight gradients (+ bias gradients) as ONE grid of 36 + 108 + 144 + 144 output tiles, every
                    # workgroup walking the whole contraction: no split-M slabs, no reduce launches (csrc/gemm_tn.hip)
                    K.gemm_tn_grouped([
                        (dy2, a["g"], self.G(Ln + "output.dense.weight"), M, H, I, beta, self.G(Ln + "output.dense.bias")),
                        (ds["dz"], a["x1"], self.G(Ln + "intermediate.dense.weight"), M, I, H, beta, self.G(Ln + "intermediate.dense.bias")),
                        (dqkv, x_in, self.G(Ln + "attention.self.query.weight"), M, 3 * H, H, beta, self.G(Ln + "attention.self.query.bias")),
                        (dy1, a["ctx"], self.G(Ln + "attention.output.dense.weight"), M, H, H, beta, self.G(Ln + "attention.output.dense.bias"))])
                else:
                    self._tn(dqkv, x_in, self.G(Ln + "attention.self.query.weight"), M, 3 * H, H, ws, beta,
                             bias=self.G(Ln + "attention.self.query.bias"))     # packed [3H, H] gradient
                self._bucket_done(NL - i)          # the layer's gradient slice is complete in this stream's order
                if use_side:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    self._side_done[i & 1] = ev
            on_side(last_wgrad)
            self._nt(dqkv, s["qkvT"], dx, M, H, 3 * H, residual=dpre)
        if use_side and not (grouped and self.TAIL_ON_SIDE):
            main.wait_stream(side)          # all layer wgrads (and their bucket hand-offs) precede the rest of backward (split-M wgrads share tn_ws)
            self._side_busy = False
        # (grouped wgrads + TAIL_ON_SIDE: the main stream does NOT wait here -- layer 0's weight gradients (170 us, issued a moment ago) and the
        # embedding tables run on the side stream underneath the embedding / region-projection backward below, which touches none of their
        # operands; the streams join once, at the end of backward)
        dpre = ws["dpre"]

        # ---- embeddings -------------------------------------------------------------------------------------
        K.layernorm_bwd(dx, ws["emb_pre"], self.P(E + "LayerNorm.weight"), ws["stat0"][0], ws["stat0"][1], dpre,
                        self.G(E + "LayerNorm.weight"), self.G(E + "LayerNorm.bias"), M, H, ln_slot(2 * NL), beta=beta, dy_drop=(p, seed, 1000),
                        defer_reduce=defer, row_map=rm)
        if rm is not None:
            # the embedding backward sums over (batch, position) in the dense [B, L] geometry (position table: a column of the batch;
            # word table: id chains in row order): hand it the dense gradient -- exact zeros on the dropped positions, as in the dense run
            dense = ws["dx_alt"]
            dense.zero_()
            K.rows_unpack(dpre, rm, M, dense, H)
            dpre = dense

        def tables_and_ln_params():
            # the tail of backward that nothing on the main stream waits for: dgamma / dbeta of the 2 * layers + 1 LayerNorms (one launch,
            # slot order = table order) and the three embedding tables (five launches); their gradient slice is announced from here
            if defer:
                K.layernorm_bwd_reduce_batched(ws["ln_slots"], self._ln_table(), 2 * NL + 1, M, H, beta=beta)
            K.embed_bwd(dpre, input_ids, token_type_ids, ws["vis_h"], ws["vispe_h"], self.G(E + "word_embeddings.weight"),
                        self.G(E + "position_embeddings.weight"), self.G(E + "token_type_embeddings.weight"), ws["d_vis_h"], ws["d_vispe_h"],
                        ws["acc32"], B, L, Nv, H, V, cfg.type_vocab_size, drop_p=p, seed=seed, vis_stream=1001, vispe_stream=1002,
                        region_mask=pt[0]["rmask"] if pt is not None else None, parts=2)
            self._bucket_done(NL + 1)           # position / type / word embedding tables (tied decoder wgrad + embedding backward) are final

        # region rows first (the region-projection dgrad / wgrads below wait for d_vis_h / d_vispe_h only); the tables and the LayerNorm
        # parameter sums run on the side stream underneath them (round 5: six small launches, ~140 us, off the critical path)
        K.embed_bwd(dpre, input_ids, token_type_ids, ws["vis_h"], ws["vispe_h"], self.G(E + "word_embeddings.weight"),
                    self.G(E + "position_embeddings.weight"), self.G(E + "token_type_embeddings.weight"), ws["d_vis_h"], ws["d_vispe_h"],
                    ws["acc32"], B, L, Nv, H, V, cfg.type_vocab_size, drop_p=p, seed=seed, vis_stream=1001, vispe_stream=1002,
                    region_mask=pt[0]["rmask"] if pt is not None else None, parts=1)
        if use_side and self.TAIL_ON_SIDE:
            self._side_busy = True
            on_side(tables_and_ln_params)
        else:
            tables_and_ln_params()
        # vis_pe_embed: Linear(1607, H) -- wgrad into the padded shadow, then crop-accumulate
        # vis_embed: Linear(2048,2048)+ReLU -> Linear(2048,H)+ReLU+Dropout
        self._nt(ws["d_vis_h"], sh["v2T"], ws["dz1v"], Mv, 2048, H, mul_src=ws["h1"], mul_mode=K.MUL_RELU_MASK)
        if grouped and Mv >= 2048:
            # the three region-projection wgrads (78 + 96 + 256 output tiles, contraction over B x 100 region rows) as one grouped launch
            K.gemm_tn_grouped([
                (ws["dz1v"], img, self.G("vis_embed.0.weight"), Mv, 2048, 2048, beta, self.G("vis_embed.0.bias")),
                (ws["d_vis_h"], ws["h1"], self.G("vis_embed.2.weight"), Mv, H, 2048, beta, self.G("vis_embed.2.bias")),
                (ws["d_vispe_h"], ws["vpe_in"], ws["dwpe_pad"], Mv, H, PE_PAD, 0, None)])
        else:
            if use_side and self._side_busy:
                # these split-M launches use ws["tn_ws"] on the MAIN stream; the head wgrads deferred to the side stream use the same scratch and
                # with <= 2 layers no per-layer wait orders the two: join first (small batches only -- Mv < 2048)
                main.wait_stream(side)
            self._tn(ws["d_vispe_h"], ws["vpe_in"], ws["dwpe_pad"], Mv, H, PE_PAD, ws, 0)
            self._tn(ws["d_vis_h"], ws["h1"], self.G("vis_embed.2.weight"), Mv, H, 2048, ws, beta, bias=self.G("vis_embed.2.bias"))
            self._tn(ws["dz1v"], img, self.G("vis_embed.0.weight"), Mv, 2048, 2048, ws, beta, bias=self.G("vis_embed.0.bias"))
        K.copy2d(ws["dwpe_pad"], PE_PAD, False, self.G("vis_pe_embed.0.weight"), PE_DIM, H, PE_DIM, PE_DIM, beta=beta)
        K.colsum(ws["d_vispe_h"], self.G("vis_pe_embed.0.bias"), Mv, H, beta=beta, workspace=ws["cs_ws"])
        if use_side and self._side_busy:
            # the streams join HERE, in front of the last hand-off: a reducer may coalesce the embedding tables (written on the side stream)
            # with the region projections into one bucket, and that bucket's collective is ordered behind the stream that announces it
            main.wait_stream(side)
            self._side_busy = False
        self._bucket_done(NL + 2)
        self.grads_dirty = True
        if self.post_backward_hook is not None:
            self.post_backward_hook()


def st_labels(st):
    return st.task_labels
