"""TEST INFRASTRUCTURE ONLY -- loads the *unmodified* reference implementation from
/root/reference (read-only) so that the oracle restatement in ``oracle/vlp_oracle.py`` can be
pinned against it and golden vectors can be generated (``oracle/make_golden.py``).

Nothing in the product (``vlp_amd/``) may import this module.  /root/reference does not exist on
the GPU box, so nothing that runs there (``-m gpu`` tests, smoke(), bench.py) may call
``load_reference()`` either; callers must check ``reference_available()`` first.

Recipe (no reference file is modified or copied):
  1. stub the third-party imports that are absent here (boto3/botocore, torch._six),
  2. register a bare namespace package ``pytorch_pretrained_bert`` whose __path__ points into the
     reference tree so that the package __init__ (which imports apex) is NOT executed,
  3. load file_utils, loss, modeling, optimization by file location,
  4. run model constructors from a temp CWD holding synthetic detectron_weights/fc7_{w,b}.pkl
     (modeling.py:1008-1014 reads them relative to the CWD).
"""
import collections.abc
import contextlib
import importlib.util
import os
import pickle
import sys
import tempfile
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("VLP_REFERENCE_ROOT", "/root/reference")
_PKG = "pytorch_pretrained_bert"
_cache = {}


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, _PKG, "modeling.py"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    """Returns a namespace with .modeling, .optimization, .loss (the reference's own modules)."""
    if "ns" in _cache:
        return _cache["ns"]
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    _stub("boto3")
    _stub("botocore")
    _stub("botocore.exceptions", ClientError=Exception)
    import torch
    if not hasattr(torch, "_six"):
        six = _stub("torch._six", container_abcs=collections.abc, string_classes=(str,), int_classes=(int,))
        torch._six = six
    pkg_dir = os.path.join(REFERENCE_ROOT, _PKG)
    saved_pkg = {k: v for k, v in sys.modules.items() if k == _PKG or k.startswith(_PKG + ".")}
    for k in saved_pkg:
        del sys.modules[k]
    pkg = types.ModuleType(_PKG)
    pkg.__path__ = [pkg_dir]
    sys.modules[_PKG] = pkg
    mods = {}
    try:
        for name in ("file_utils", "loss", "modeling", "optimization"):
            spec = importlib.util.spec_from_file_location(_PKG + "." + name, os.path.join(pkg_dir, name + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[_PKG + "." + name] = mod
            with open(os.devnull, "w") as dn, contextlib.redirect_stdout(dn):
                spec.loader.exec_module(mod)   # modeling.py prints an apex hint at import
            mods[name] = mod
    finally:
        # do not leave the reference registered under the package name: the product ships its own
        # drop-in ``pytorch_pretrained_bert`` and tests import both in one process.
        for k in list(sys.modules):
            if k == _PKG or k.startswith(_PKG + "."):
                del sys.modules[k]
        sys.modules.update(saved_pkg)
    ns = types.SimpleNamespace(**mods)
    _cache["ns"] = ns
    return ns


@contextlib.contextmanager
def detectron_cwd(seed=1234, std=0.02):
    """chdir into a temp dir that holds synthetic fc7 pickles (f32 [2048,2048], f32 [2048])."""
    old = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "detectron_weights"))
        rng = np.random.RandomState(seed)
        w = (rng.standard_normal((2048, 2048)) * std).astype(np.float32)
        b = (rng.standard_normal((2048,)) * std).astype(np.float32)
        with open(os.path.join(d, "detectron_weights", "fc7_w.pkl"), "wb") as f:
            pickle.dump(w, f)
        with open(os.path.join(d, "detectron_weights", "fc7_b.pkl"), "wb") as f:
            pickle.dump(b, f)
        os.chdir(d)
        try:
            yield d
        finally:
            os.chdir(old)


def build_reference_model(cfg_kwargs, tasks="img2txt", seed=0, drop_prob=0.0, decoder=False, **dec_kwargs):
    """Instantiate the reference BertForPreTrainingLossMask (or BertForSeq2SeqDecoder) directly
    (bypasses from_pretrained's S3 download), fp32, eval-mode dropout controlled by drop_prob."""
    import torch
    ref = load_reference()
    kw = dict(cfg_kwargs)
    kw.setdefault("type_vocab_size", 6)
    kw["hidden_dropout_prob"] = drop_prob
    kw["attention_probs_dropout_prob"] = drop_prob
    vocab = kw.pop("vocab_size")
    config = ref.modeling.BertConfig(vocab, **kw)
    torch.manual_seed(seed)
    with detectron_cwd():
        if decoder:
            model = ref.modeling.BertForSeq2SeqDecoder(config, enable_butd=True, len_vis_input=100, **dec_kwargs)
        else:
            model = ref.modeling.BertForPreTrainingLossMask(config, num_labels=2, enable_butd=True,
                                                            len_vis_input=100, tasks=tasks)
    return model


# ----------------------------------------------------------------------------------------------
# the reference's data pipeline (vlp/seq2seq_loader.py), for the N2 row: on-device input preparation
# ----------------------------------------------------------------------------------------------
H5_REGISTRY = {}          # fake file name -> {dataset key: numpy array}; served by the stub h5py below


class _FakeH5File(object):
    def __init__(self, name, mode="r"):
        if name not in H5_REGISTRY:
            raise IOError("fake h5py: no such file %r" % name)
        self._d = H5_REGISTRY[name]

    def __enter__(self):
        return self._d

    def __exit__(self, *a):
        return False


def load_reference_loader():
    """Returns the reference's own vlp.seq2seq_loader module (unmodified).  torchvision (image transforms, unused with region
    features) is stubbed; h5py is replaced by an in-memory fake so that Preprocess4Seq2seq.__call__ (seq2seq_loader.py:229-359)
    reads arrays registered in H5_REGISTRY instead of the dataset's .h5 files."""
    if "loader" in _cache:
        return _cache["loader"]
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    tv = _stub("torchvision")
    tr = _stub("torchvision.transforms", Resize=lambda *a, **k: None, RandomCrop=lambda *a, **k: None, ToTensor=lambda *a, **k: None,
               Normalize=lambda *a, **k: None)
    tv.transforms = tr
    real_h5 = sys.modules.get("h5py")
    sys.modules["h5py"] = types.SimpleNamespace(File=_FakeH5File)
    saved = {k: v for k, v in sys.modules.items() if k == "vlp" or k.startswith("vlp.")}
    for k in saved:
        del sys.modules[k]
    pkg = types.ModuleType("vlp")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "vlp")]
    sys.modules["vlp"] = pkg
    try:
        mods = {}
        for name in ("loader_utils", "seq2seq_loader"):
            spec = importlib.util.spec_from_file_location("vlp." + name, os.path.join(REFERENCE_ROOT, "vlp", name + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules["vlp." + name] = mod
            spec.loader.exec_module(mod)
            mods[name] = mod
    finally:
        for k in list(sys.modules):
            if k == "vlp" or k.startswith("vlp."):
                del sys.modules[k]
        sys.modules.update(saved)
        if real_h5 is not None:
            sys.modules["h5py"] = real_h5
        else:
            del sys.modules["h5py"]
    _cache["loader"] = mods["seq2seq_loader"]
    return _cache["loader"]
