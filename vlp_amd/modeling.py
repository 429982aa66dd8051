"""Drop-in for `pytorch_pretrained_bert.modeling` of LuoweiZhou/VLP on MI355X.

Same public surface as the reference file (class names, constructor / forward signatures, submodule
tree and therefore state_dict keys, `from_pretrained` keyword handling -- citations below are into
pytorch_pretrained_bert/modeling.py), but the module tree is only a *container of parameters*: all
arithmetic of `BertForPreTrainingLossMask.forward` runs as hand-written HIP kernels through
libvlp_hip.so, sequenced by vlp_amd.engine.Engine (one fused forward/backward, no per-op autograd graph).

Not implemented (raise loudly; see DESIGN.md "out of scope"): fp32 execution, `enable_butd=False`
(the reference asserts it is True, run_img2txt_dist.py:199), `relax_projection`, label smoothing, and the dead HF heads
(:878-978, :1497-1966).  `mask_image_regions` / `vis_pretext_loss` (:1049-1056, 1113-1131) and the pooler they use are built.
"""
import copy
import json
import logging
import math
import os
import pickle

import torch
from torch import nn

from . import _lib as K
from .engine import Engine

logger = logging.getLogger(__name__)

CONFIG_NAME = "bert_config.json"
WEIGHTS_NAME = "pytorch_model.bin"
PRETRAINED_MODEL_ARCHIVE_MAP = {   # names the reference resolves over the network (:49-57); offline they must be local dirs
    "bert-base-uncased": None, "bert-large-uncased": None, "bert-base-cased": None, "bert-large-cased": None,
    "bert-base-multilingual-uncased": None, "bert-base-multilingual-cased": None, "bert-base-chinese": None,
}


class BertConfig(object):
    """Same fields and helpers as the reference BertConfig (:77-171)."""

    def __init__(self, vocab_size_or_config_json_file, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, relax_projection=0, initializer_range=0.02, task_idx=None,
                 fp32_embedding=False, label_smoothing=None):
        if isinstance(vocab_size_or_config_json_file, str):
            with open(vocab_size_or_config_json_file, "r", encoding="utf-8") as reader:
                for key, value in json.loads(reader.read()).items():
                    self.__dict__[key] = value
        elif isinstance(vocab_size_or_config_json_file, int):
            self.vocab_size = vocab_size_or_config_json_file
            self.hidden_size = hidden_size
            self.num_hidden_layers = num_hidden_layers
            self.num_attention_heads = num_attention_heads
            self.hidden_act = hidden_act
            self.intermediate_size = intermediate_size
            self.hidden_dropout_prob = hidden_dropout_prob
            self.attention_probs_dropout_prob = attention_probs_dropout_prob
            self.max_position_embeddings = max_position_embeddings
            self.type_vocab_size = type_vocab_size
            self.relax_projection = relax_projection
            self.initializer_range = initializer_range
            self.task_idx = task_idx
            self.fp32_embedding = fp32_embedding
            self.label_smoothing = label_smoothing
        else:
            raise ValueError("First argument must be either a vocabulary size (int) or the path to a pretrained model config file (str)")

    @classmethod
    def from_dict(cls, json_object):
        config = BertConfig(vocab_size_or_config_json_file=-1)
        for key, value in json_object.items():
            config.__dict__[key] = value
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as reader:
            return cls.from_dict(json.loads(reader.read()))

    def __repr__(self):
        return str(self.to_json_string())

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"


def _no_direct_forward(self, *a, **k):
    raise NotImplementedError("%s is a parameter container in vlp_amd: the fused HIP path is entered through "
                              "BertForPreTrainingLossMask.forward / BertForSeq2SeqDecoder.forward" % type(self).__name__)


class BertLayerNorm(nn.Module):
    """Parameters of the TF-style LayerNorm (:174-192); evaluated by vlp_layernorm_fwd/bwd."""

    def __init__(self, hidden_size, eps=1e-5):
        super(BertLayerNorm, self).__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        """Stand-alone use (any [*, H] fp16 device tensor) through the C ABI."""
        H = x.shape[-1]
        x2 = x.reshape(-1, H).contiguous()
        y = torch.empty_like(x2)
        K.layernorm_fwd(x2, self.weight.data, self.bias.data, y, x2.shape[0], H, eps=self.variance_epsilon)
        return y.view(x.shape)


class BertEmbeddings(nn.Module):   # :195-241
    def __init__(self, config):
        super(BertEmbeddings, self).__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.fp32_embedding = getattr(config, "fp32_embedding", False)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-5)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
    forward = _no_direct_forward


class BertSelfAttention(nn.Module):   # :244-303
    def __init__(self, config):
        super(BertSelfAttention, self).__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
    forward = _no_direct_forward


class BertSelfOutput(nn.Module):   # :306-317
    def __init__(self, config):
        super(BertSelfOutput, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-5)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
    forward = _no_direct_forward


class BertAttention(nn.Module):   # :320-330
    def __init__(self, config):
        super(BertAttention, self).__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)
    forward = _no_direct_forward


class BertIntermediate(nn.Module):   # :333-343
    def __init__(self, config):
        super(BertIntermediate, self).__init__()
        if config.hidden_act != "gelu":
            raise NotImplementedError("vlp_amd fuses the erf-GELU of the reference configs; hidden_act=%r is not supported" % (config.hidden_act,))
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
    forward = _no_direct_forward


class BertOutput(nn.Module):   # :346-357
    def __init__(self, config):
        super(BertOutput, self).__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-5)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
    forward = _no_direct_forward


class BertLayer(nn.Module):   # :360-372
    def __init__(self, config):
        super(BertLayer, self).__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
    forward = _no_direct_forward


class BertEncoder(nn.Module):   # :375-402
    def __init__(self, config):
        super(BertEncoder, self).__init__()
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])
    forward = _no_direct_forward


class BertPooler(nn.Module):   # :405-417
    def __init__(self, config):
        super(BertPooler, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()
    forward = _no_direct_forward


class BertPredictionHeadTransform(nn.Module):   # :420-435
    def __init__(self, config):
        super(BertPredictionHeadTransform, self).__init__()
        if getattr(config, "relax_projection", 0) > 1:
            raise NotImplementedError("relax_projection > 1 is not supported by vlp_amd (off in every reference config)")
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-5)
    forward = _no_direct_forward


class BertLMPredictionHead(nn.Module):   # :438-482
    def __init__(self, config, bert_model_embedding_weights):
        super(BertLMPredictionHead, self).__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(bert_model_embedding_weights.size(1), bert_model_embedding_weights.size(0), bias=False)
        self.decoder.weight = bert_model_embedding_weights      # tied (:445-448)
        self.bias = nn.Parameter(torch.zeros(bert_model_embedding_weights.size(0)))
        self.relax_projection = 0
        self.fp32_embedding = getattr(config, "fp32_embedding", False)
    forward = _no_direct_forward


class BertPreTrainingHeads(nn.Module):   # :506-520
    def __init__(self, config, bert_model_embedding_weights, num_labels=2):
        super(BertPreTrainingHeads, self).__init__()
        self.predictions = BertLMPredictionHead(config, bert_model_embedding_weights)
    forward = _no_direct_forward


def load_checkpoint_state(model, state_dict):
    """The checkpoint remapping of the reference's from_pretrained (:651-752), shared by `from_pretrained` and the train
    script: TF-era gamma/beta names, segment table 2 -> 6 rows, position-table tiling, size-mismatch errors; sets
    `model.missing_keys`.  Anything whose shape does not fit after the remapping raises (never silently skipped)."""
    config = model.config
    state_dict = dict(state_dict)
    # TF-era names (:651-663)
    for key in list(state_dict.keys()):
        new_key = key.replace("gamma", "weight").replace("beta", "bias")
        if new_key != key:
            state_dict[new_key] = state_dict.pop(key)
    H = config.hidden_size
    # segment table 2 -> 6 rows: rows 2,3,4 start from row 0 and row 5 from row 1 (:665-683)
    k = "bert.embeddings.token_type_embeddings.weight"
    if k in state_dict and state_dict[k].shape[0] != config.type_vocab_size:
        old = state_dict[k]
        if config.type_vocab_size > old.shape[0]:
            new = old.new_zeros(config.type_vocab_size, H)
            new.normal_(mean=0.0, std=config.initializer_range)
            new[:old.shape[0]] = old
            if config.type_vocab_size >= 6:
                new[2], new[3], new[4], new[5] = old[0], old[0], old[0], old[1]
            state_dict[k] = new
        else:
            state_dict[k] = old[:config.type_vocab_size]
    # position table: tile the learned rows when it grows (:685-702)
    k = "bert.embeddings.position_embeddings.weight"
    if k in state_dict and state_dict[k].shape[0] != config.max_position_embeddings:
        old = state_dict[k]
        n_old, n_new = old.shape[0], config.max_position_embeddings
        if n_new > n_old:
            reps = int(math.ceil(n_new / float(n_old)))
            state_dict[k] = old.repeat(reps, 1)[:n_new].clone()
        else:
            state_dict[k] = old[:n_new]
    k = "cls.predictions.transform.dense.weight"
    if k in state_dict and state_dict[k].shape[0] != H:
        raise NotImplementedError("checkpoint uses relax_projection; not supported by vlp_amd")
    own = model.state_dict()
    # a bare BertModel reads the 'bert.'-prefixed entries of a task checkpoint (:751)
    strip = "" if hasattr(model, "bert") else "bert."
    missing, unexpected, load = [], [], {}
    for key, value in state_dict.items():
        if strip and not key.startswith(strip):
            unexpected.append(key)
            continue
        full = key[len(strip):]
        if full in own:
            if tuple(own[full].shape) != tuple(value.shape):
                raise RuntimeError("size mismatch for %s: checkpoint %s vs model %s" % (full, tuple(value.shape), tuple(own[full].shape)))
            load[full] = value
        else:
            unexpected.append(key)
    for key in own:
        if key not in load:
            missing.append(key)
    nn.Module.load_state_dict(model, load, strict=False)
    model.missing_keys = missing
    if missing:
        logger.info("Weights of {} not initialized from pretrained model: {}".format(model.__class__.__name__, missing))
    if unexpected:
        logger.info("Weights from pretrained model not used in {}: {}".format(model.__class__.__name__, unexpected))
    return model


class PreTrainedBertModel(nn.Module):
    """Weight init and `from_pretrained` (:523-764)."""

    def __init__(self, config, *inputs, **kwargs):
        super(PreTrainedBertModel, self).__init__()
        if not isinstance(config, BertConfig):
            raise ValueError("Parameter config in `{}(config)` should be an instance of class `BertConfig`.".format(self.__class__.__name__))
        self.config = config

    def init_bert_weights(self, module):   # :539-551
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    # parameters live in flat device buffers once packed: any storage-changing _apply must unlink them
    def _apply(self, fn, *a, **k):
        eng = self.__dict__.get("_engine")
        out = super(PreTrainedBertModel, self)._apply(fn, *a, **k)
        if eng is not None:
            eng.invalidate()
        return out

    def __deepcopy__(self, memo):
        # deepcopy (run_img2txt_dist.py:598 `copy.deepcopy(model_to_save).cpu()`) must not clone workspaces
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_engine":
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new.__dict__["_engine"] = Engine(new) if "_engine" in self.__dict__ else None
        for p in new.parameters():
            p.grad = None
            if hasattr(p, "_vlp_engine"):
                del p._vlp_engine
            if new.__dict__["_engine"] is not None:
                p._vlp_owner = new.__dict__["_engine"]
        return new

    def state_dict(self, *args, **kwargs):
        """Same keys as the reference; tensors are detached clones so a checkpoint never drags the flat
        parameter buffer along."""
        eng = self.__dict__.get("engine") or getattr(self, "engine", None)
        if eng is not None and getattr(eng, "packed", False):
            eng.wait_params()            # a pipelined optimizer step may still be writing the flat parameter buffers
        sd = super(PreTrainedBertModel, self).state_dict(*args, **kwargs)
        if args or kwargs.get("destination") is not None:
            return sd            # nested call from a parent module: the top-level call clones once
        for k in list(sd.keys()):
            sd[k] = sd[k].detach().clone()
        return sd

    @classmethod
    def from_pretrained(cls, pretrained_model_name, state_dict=None, cache_dir=None, *inputs, **kwargs):
        """Same contract as the reference (:554-764): `pretrained_model_name` is a directory holding
        bert_config.json (+ pytorch_model.bin unless `state_dict` is given; `state_dict={}` means random
        init, run_img2txt_dist.py:320-323).  Model *names* need the network in the reference and are
        rejected here."""
        if pretrained_model_name in PRETRAINED_MODEL_ARCHIVE_MAP and not os.path.isdir(pretrained_model_name):
            config_path = kwargs.get("config_path")
            if not config_path:
                raise EnvironmentError("'%s' would be downloaded by the reference; offline, pass a local model directory "
                                       "(bert_config.json [+ pytorch_model.bin]) or config_path=..." % pretrained_model_name)
            serialization_dir = None
        else:
            serialization_dir = pretrained_model_name
            if not os.path.isdir(serialization_dir):
                logger.error("Model directory '%s' not found", pretrained_model_name)
                return None
        config_file = kwargs.get("config_path") or os.path.join(serialization_dir, CONFIG_NAME)
        config = BertConfig.from_json_file(config_file)
        # keyword overrides (:615-636)
        if "type_vocab_size" in kwargs:
            config.type_vocab_size = kwargs["type_vocab_size"]
        for key in ("relax_projection", "task_idx", "max_position_embeddings", "fp32_embedding", "label_smoothing"):
            if kwargs.get(key):
                setattr(config, key, kwargs[key])
        if "drop_prob" in kwargs:
            config.attention_probs_dropout_prob = kwargs["drop_prob"]
            config.hidden_dropout_prob = kwargs["drop_prob"]
        for key in ("config_path", "type_vocab_size", "relax_projection", "task_idx", "max_position_embeddings", "fp32_embedding",
                    "label_smoothing", "drop_prob"):
            kwargs.pop(key, None)
        for key in ("relax_projection", "task_idx", "fp32_embedding", "label_smoothing"):
            if not hasattr(config, key):
                setattr(config, key, None if key in ("task_idx", "label_smoothing") else 0)
        logger.info("Model config {}".format(config))
        model = cls(config, *inputs, **kwargs)
        if state_dict is None:
            if serialization_dir is None:
                raise EnvironmentError("no weights: pass state_dict (or {} for random init)")
            state_dict = torch.load(os.path.join(serialization_dir, WEIGHTS_NAME), map_location="cpu")
        load_checkpoint_state(model, state_dict)
        return model


class BertModel(PreTrainedBertModel):
    """Embeddings + encoder + pooler parameters (:767-849).  As a sub-module of the task models it is a
    container; the fused path owns its arithmetic."""

    def __init__(self, config):
        super(BertModel, self).__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.pooler = BertPooler(config)
        self.apply(self.init_bert_weights)
    forward = _no_direct_forward


class BertModelIncr(BertModel):   # :852-875
    def __init__(self, config):
        super(BertModelIncr, self).__init__(config)


def _region_embedders(module, config, enable_butd, allow_random_fc7):
    """vis_embed / vis_pe_embed exactly as the reference builds them (:1002-1018)."""
    if not enable_butd:
        raise NotImplementedError("vlp_amd supports region features only (enable_butd=True); the reference itself asserts this "
                                  "(run_img2txt_dist.py:199 'featmap attn deprecated')")
    module.vis_embed = nn.Sequential(nn.Linear(2048, 2048), nn.ReLU(), nn.Linear(2048, config.hidden_size), nn.ReLU(),
                                     nn.Dropout(config.hidden_dropout_prob))
    module.vis_pe_embed = nn.Sequential(nn.Linear(6 + 1601, config.hidden_size), nn.ReLU(), nn.Dropout(config.hidden_dropout_prob))
    return allow_random_fc7


def _load_fc7(module, allow_random):
    """Detectron fc7 weights are read relative to the CWD like the reference (:1008-1014)."""
    try:
        with open("detectron_weights/fc7_w.pkl", "rb") as f:
            w = torch.from_numpy(pickle.load(f))
        with open("detectron_weights/fc7_b.pkl", "rb") as f:
            b = torch.from_numpy(pickle.load(f))
        module.vis_embed[0].weight.data.copy_(w)
        module.vis_embed[0].bias.data.copy_(b)
    except Exception:
        if allow_random or os.environ.get("VLP_ALLOW_RANDOM_FC7") == "1":
            return
        raise Exception("Cannot find Detectron fc7 weights! Download from https://dl.fbaipublicfiles.com/ActivityNet-Entities/"
                        "ActivityNet-Entities/detectron_weights.tar.gz and uncompress under the code root directory "
                        "(or pass allow_random_fc7=True / VLP_ALLOW_RANDOM_FC7=1 for synthetic runs).")


class _LossFn(torch.autograd.Function):
    """Hands the fused backward to autograd: `loss.backward()` (run_img2txt_dist.py:571-575) reaches
    Engine.backward with the upstream scalar gradient (which carries the fp16 loss scale).  With mask_image_regions the forward has two
    live losses (task loss, vis_pretext_loss; the train loop sums them, :531): both are outputs of ONE node, so a single backward call
    receives both upstream gradients."""

    @staticmethod
    def forward(ctx, anchor, loss, pretext, engine, state, task):
        ctx.engine, ctx.state, ctx.task = engine, state, task
        ctx.has_pretext = pretext is not None
        if pretext is None:
            return loss.clone()
        return loss.clone(), pretext.clone()

    @staticmethod
    def backward(ctx, grad, grad_pt=None):
        g = grad.detach().to(torch.float32).reshape(1).contiguous()
        gp = grad_pt.detach().to(torch.float32).reshape(1).contiguous() if ctx.has_pretext else None
        ctx.engine.backward(ctx.state, g, ctx.task, g_pretext=gp)
        return None, None, None, None, None, None


class BertForPreTrainingLossMask(PreTrainedBertModel):
    """The VLP training / VQA model (:982-1143) on the fused HIP path."""

    def __init__(self, config, num_labels=2, enable_butd=False, len_vis_input=49, tasks="img2txt", allow_random_fc7=False):
        super(BertForPreTrainingLossMask, self).__init__(config)
        self.bert = BertModel(config)
        self.cls = BertPreTrainingHeads(config, self.bert.embeddings.word_embeddings.weight, num_labels=num_labels)
        self.apply(self.init_bert_weights)
        self.num_labels = num_labels
        self.len_vis_input = len_vis_input
        self.enable_butd = enable_butd
        if getattr(config, "label_smoothing", None):
            raise NotImplementedError("label smoothing (loss.py) is off in every reference config and not implemented in vlp_amd")
        self.crit_mask_lm_smoothed = None
        _region_embedders(self, config, enable_butd, allow_random_fc7)
        _load_fc7(self, allow_random_fc7)
        self.tasks = tasks
        self.num_answers = 3129          # hard-coded in the reference (:1029)
        if tasks == "vqa2":
            self.ans_classifier = nn.Sequential(nn.Linear(config.hidden_size, config.hidden_size * 2), nn.ReLU(),
                                                nn.Linear(config.hidden_size * 2, self.num_answers))
        self.__dict__["_engine"] = Engine(self)
        for prm in self.parameters():          # lets optimizers / DDP find the engine before the first forward packs
            prm._vlp_owner = self.__dict__["_engine"]

    @property
    def engine(self):
        return self.__dict__["_engine"]

    def forward(self, vis_feats, vis_pe, input_ids, token_type_ids=None, attention_mask=None, masked_lm_labels=None, ans_labels=None,
                next_sentence_label=None, masked_pos=None, masked_weights=None, task_idx=None, vis_masked_pos=[],
                mask_image_regions=False, drop_worst_ratio=0.2, vqa_inference=False):
        eng = self.engine
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        train = self.training

        if vqa_inference:                                              # :1039-1047 (returns before the region masking of :1049)
            assert ans_labels is None
            st = eng.forward(vis_feats, vis_pe, input_ids, token_type_ids, attention_mask, None, False, False, True)
            logits = eng.vqa_logits(st)
            self.last_vqa_logits = logits
            return torch.max(logits[:, 1:].float(), -1)[1] + 1
        if masked_lm_labels is None or next_sentence_label is None:    # :1062-1063
            raise NotImplementedError
        vmp = None
        if mask_image_regions:                                         # :1049-1056, 1113-1131
            if not torch.is_tensor(vis_masked_pos) or vis_masked_pos.dim() != 2 or vis_masked_pos.shape[1] == 0:
                raise ValueError("mask_image_regions=True needs vis_masked_pos [B, n_masked] (seq2seq_loader.py:267-269)")
            vmp = vis_masked_pos
        is_vqa = self.tasks == "vqa2"
        if is_vqa:
            assert ans_labels is not None
        want_mlm = (not is_vqa) and masked_pos is not None and masked_pos.numel() > 0
        st = eng.forward(vis_feats, vis_pe, input_ids, token_type_ids, attention_mask, masked_pos, train, want_mlm, is_vqa, vis_masked_pos=vmp)
        # the reference's `.new(1).fill_(0)` placeholders of the losses a task does not have (:1096-1098, 1133): ONE read-only [1] zero per engine,
        # handed out by identity (no fill / clone launches per step; vlp_amd.run_img2txt_dist.train_step skips adding it) -- do not modify it in place
        zero1 = eng.zero_placeholder(input_ids.device)
        need_grad = torch.is_grad_enabled()
        raw_pt = eng.pretext_loss(st) if vmp is not None else None
        if vmp is not None:
            self.last_pooled_output = eng.pooled_output(st)

        def live(raw, task):
            """(task loss, pretext loss) as autograd outputs of one node; the pretext placeholder is the reference's [1] zero (:1133)."""
            if not need_grad:
                return raw.clone(), (raw_pt.clone().reshape(()) if raw_pt is not None else zero1)
            if raw_pt is None:
                return _LossFn.apply(eng._anchor, raw, None, eng, st, task), zero1
            a, b = _LossFn.apply(eng._anchor, raw, raw_pt, eng, st, task)
            return a, b.reshape(())                                     # :1131: a 0-dim mean

        if is_vqa:
            raw = eng.vqa_loss(st, ans_labels)
            self.last_vqa_logits = eng.vqa_logits(st)
            loss, pt_loss = live(raw, "vqa2")
            return zero1, pt_loss, loss.reshape(())                     # :1141 shapes ([1], [1] | [], [])
        if not want_mlm:
            if raw_pt is None:
                return zero1, zero1, zero1                              # :1096-1098
            loss, pt_loss = live(zero1.clone(), "img2txt")              # empty masked_pos: only the pretext loss is live
            return loss, pt_loss, zero1
        raw = eng.mlm_loss(st, masked_lm_labels, masked_weights, drop_worst_ratio)
        self.last_mlm_logits = eng.mlm_logits(st)
        loss, pt_loss = live(raw, "img2txt")
        return loss.reshape(()), pt_loss, zero1                         # :1143 shapes ([], [1] | [], [1])


class BertForSeq2SeqDecoder(PreTrainedBertModel):
    """Incremental caption decoder (:1147-1494): same parameter tree as BertForPreTrainingLossMask (checkpoint compatible);
    greedy decoding (:1189-1253) and beam search (:1255-1494) run on the HIP engine with per-layer K/V caches
    (Engine.decode_greedy / Engine.decode_beam); the host keeps only what the reference keeps on the host (n-gram blocking
    over python lists, back-tracking of the frames).  sample_mode='sample' draws with the library's counter-based RNG."""
    tasks = "img2txt"

    def __init__(self, config, mask_word_id=0, num_labels=2, search_beam_size=1, length_penalty=1.0, eos_id=0,
                 forbid_duplicate_ngrams=False, forbid_ignore_set=None, ngram_size=3, min_len=0, enable_butd=False, len_vis_input=49,
                 allow_random_fc7=True):
        super(BertForSeq2SeqDecoder, self).__init__(config)
        self.bert = BertModelIncr(config)
        self.cls = BertPreTrainingHeads(config, self.bert.embeddings.word_embeddings.weight, num_labels=num_labels)
        self.apply(self.init_bert_weights)
        self.mask_word_id, self.num_labels, self.len_vis_input = mask_word_id, num_labels, len_vis_input
        self.search_beam_size, self.length_penalty, self.eos_id = search_beam_size, length_penalty, eos_id
        self.forbid_duplicate_ngrams, self.forbid_ignore_set, self.ngram_size, self.min_len = forbid_duplicate_ngrams, forbid_ignore_set, ngram_size, min_len
        _region_embedders(self, config, enable_butd, allow_random_fc7)
        self.__dict__["_engine"] = Engine(self)
        for prm in self.parameters():
            prm._vlp_owner = self.__dict__["_engine"]

    @property
    def engine(self):
        return self.__dict__["_engine"]

    def forward(self, vis_feats, vis_pe, input_ids, token_type_ids, position_ids, attention_mask, task_idx=None, sample_mode="greedy"):
        """Returns (output_ids [B, n], output_probs [B, n]) with n = token_type_ids.shape[1] - input_ids.shape[1], as :1253.
        output_probs are the maximal prediction scores (logits), exactly what the reference returns in greedy mode (:1228)."""
        if not input_ids.is_cuda:
            raise RuntimeError("vlp_amd: the decoder runs on the HIP engine only (inputs must be on the GPU); there is no CPU path")
        if self.search_beam_size > 1:
            with torch.no_grad():
                return self.beam_search(vis_feats, vis_pe, input_ids, token_type_ids, position_ids, attention_mask, task_idx=task_idx)
        if sample_mode not in ("greedy", "sample"):
            raise NotImplementedError("sample_mode=%r (the reference knows 'greedy' and 'sample', modeling.py:1227-1237)" % (sample_mode,))
        with torch.no_grad():
            return self.engine.decode_greedy(vis_feats, vis_pe, input_ids, token_type_ids, position_ids, attention_mask, self.mask_word_id,
                                             sample=(sample_mode == "sample"))

    # ---- beam search: host side (what the reference also does on the host) -------------------------------
    def _ngram_blocker(self, batch_size, vocab_size):
        """Stateful callback for Engine.decode_beam: tracks the partial hypotheses (:1367-1384) and returns the uint8 [B*K, V] mask
        of words that would repeat an n-gram (:1386-1430), or None."""
        import numpy as np
        K, n, ignore = self.search_beam_size, self.ngram_size, self.forbid_ignore_set
        state = {"seqs": None}

        def repeats(seq):
            if len(seq) < n:
                return ()
            tail = seq[len(seq) - (n - 1):]
            if ignore and any(t in ignore for t in tail):
                return ()
            hits = set()
            for i in range(len(seq) - (n - 1)):
                if seq[i:i + n - 1] == tail:
                    nxt = seq[i + n - 1]
                    if not (ignore and nxt in ignore):
                        hits.add(nxt)
            return hits

        def step(ids, back, first):
            if first:
                state["seqs"] = [[ids[b][k]] for b in range(batch_size) for k in range(K)]
            else:
                old = state["seqs"]
                state["seqs"] = [old[b * K + back[b][k]] + [ids[b][k]] for b in range(batch_size) for k in range(K)]
            seqs = state["seqs"]
            if len(seqs[0]) < n:
                return None
            hits = [repeats(sq) for sq in seqs]
            if not any(hits):
                return None
            mask = np.zeros((batch_size * K, vocab_size), dtype=np.uint8)
            for r, hs in enumerate(hits):
                for w in hs:
                    mask[r, w] = 1
            return mask
        return step

    def _backtrack(self, scores, words, back):
        """One sample's frames (lists [frames][K]) -> best hypothesis (:1446-1474): candidates are the hypotheses that emitted eos, or
        any hypothesis of the last valid frame (the first frame whose words are all eos, else the final one); rank by cumulative
        log-probability + length_penalty * length; follow the back pointers."""
        eos, lp = self.eos_id, self.length_penalty
        last = next((i for i, w in enumerate(words) if all(x == eos for x in w)), len(scores) - 1)
        best = None
        for f in range(last + 1):
            for k, w in enumerate(words[f]):
                if w == eos or f == last:
                    val = scores[f][k] + lp * (f + 1)
                    if best is None or val > best[0]:
                        best = (val, f, k)
        if best is None or best[0] == -math.inf:
            return [0]
        _, f, k = best
        out = [words[f][k]]
        while f > 0:
            k = back[f][k]
            f -= 1
            out.append(words[f][k])
        out.reverse()
        return out

    def beam_search(self, vis_feats, vis_pe, input_ids, token_type_ids, position_ids, attention_mask, task_idx=None):
        """Returns the reference's traces dict (:1436-1494): 'pred_seq' [B, L] (0 padded), 'scores' f32 / 'wids' / 'ptrs' [B, L, K]
        (frames padded with zeros to the output length L), on the input device."""
        B, out_len = input_ids.shape[0], token_type_ids.shape[1]
        K = self.search_beam_size
        forbid_fn = self._ngram_blocker(B, self.config.vocab_size) if self.forbid_duplicate_ngrams else None
        tot, wids, ptrs = self.engine.decode_beam(vis_feats, vis_pe, input_ids, token_type_ids, position_ids, attention_mask, self.mask_word_id,
                                                  K, self.eos_id, min_len=self.min_len, forbid_fn=forbid_fn)
        frames = tot.shape[0]
        dev = input_ids.device
        # frames stay on the device: [frames, B, K] -> [B, out_len, K], zero padded (three small launches; no per-sample host loop)
        scores = torch.zeros(B, out_len, K, dtype=torch.float32, device=dev)
        wid_o = torch.zeros(B, out_len, K, dtype=torch.long, device=dev)
        ptr_o = torch.zeros(B, out_len, K, dtype=torch.long, device=dev)
        scores[:, :frames] = tot.permute(1, 0, 2)
        wid_o[:, :frames] = wids.permute(1, 0, 2)
        ptr_o[:, :frames] = ptrs.permute(1, 0, 2)
        # best hypothesis of every sample (:1446-1474), vectorised over the batch on the host: ONE device -> host copy of the three frame tensors
        pred = self._backtrack_batch(tot.cpu().numpy(), wids.cpu().numpy(), ptrs.cpu().numpy(), out_len)
        return {"pred_seq": torch.from_numpy(pred).to(dev), "scores": scores, "wids": wid_o, "ptrs": ptr_o}

    def _backtrack_batch(self, scores, words, back, out_len):
        """_backtrack for all samples at once (numpy; arrays [frames, B, K]) -> int64 [B, out_len], 0 padded.  Same rules: candidates are the
        hypotheses that emitted eos, or any hypothesis of the last valid frame (the first frame whose K words are all eos, else the final one);
        rank by cumulative log-probability + length_penalty * (frame + 1), first maximum in (frame, beam) order; follow the back pointers."""
        import numpy as np
        F, B, K = scores.shape
        eos, lp = self.eos_id, self.length_penalty
        all_eos = (words == eos).all(axis=2)                                   # [F, B]
        last = np.where(all_eos.any(axis=0), all_eos.argmax(axis=0), F - 1)    # [B]
        f_idx = np.arange(F)[:, None, None]
        cand = (f_idx <= last[None, :, None]) & ((words == eos) | (f_idx == last[None, :, None]))
        val = np.where(cand, scores.astype(np.float64) + lp * (f_idx + 1), -np.inf)      # [F, B, K]
        flat = val.transpose(1, 0, 2).reshape(B, F * K)
        best = flat.argmax(axis=1)                                             # first maximum in (frame, beam) order, as the reference's strict `>` scan
        ok = np.isfinite(flat[np.arange(B), best])
        bf, bk = best // K, best % K
        out = np.zeros((B, out_len), dtype=np.int64)
        rows = np.arange(B)
        k = bk.copy()
        for t in range(F - 1, -1, -1):
            live = ok & (t <= bf)
            out[live, t] = words[t, rows, k][live]
            if t > 0:
                k = np.where(live, back[t, rows, k], k)
        return out
