"""Import-path compatibility for callers of the reference that are not edited at all.

    import vlp_amd.compat; vlp_amd.compat.install()

registers `pytorch_pretrained_bert` (+ `.modeling`, `.optimization`, `.optimization_fp16`) in sys.modules as aliases of the vlp_amd
modules, so that the import lines of vlp/run_img2txt_dist.py:23-30,405 and vlp/decode_img2txt.py resolve to the HIP implementation:

    from pytorch_pretrained_bert.modeling import BertForPreTrainingLossMask, BertForSeq2SeqDecoder
    from pytorch_pretrained_bert.optimization import BertAdam, warmup_linear
    from pytorch_pretrained_bert.optimization_fp16 import FP16_Optimizer_State

`misc.data_parallel` (DataParallelImbalance, run_img2txt_dist.py:30) resolves to vlp_amd.data_parallel, and, when apex is not
installed, an `apex.optimizers` module exposing `FusedAdam` is provided (run_img2txt_dist.py:406 imports it from there).
The tokenizer (`pytorch_pretrained_bert.tokenization`) is NOT provided: it is CPU-side text processing outside the hot path; keep
using the reference's file for it.  install() refuses to shadow an already imported package of that name unless force=True.
"""
import sys
import types

_ALIASES = ("modeling", "optimization", "optimization_fp16")


def install(force=False, with_apex_shim=True):
    from . import modeling, optimization, optimization_fp16
    name = "pytorch_pretrained_bert"
    if name in sys.modules and not getattr(sys.modules[name], "__vlp_amd_alias__", False) and not force:
        raise RuntimeError("a different `%s` is already imported; call install(force=True) to replace it" % name)
    pkg = types.ModuleType(name)
    pkg.__vlp_amd_alias__ = True
    pkg.__path__ = []                       # a package: submodule imports consult sys.modules first
    mods = {"modeling": modeling, "optimization": optimization, "optimization_fp16": optimization_fp16}
    for sub, mod in mods.items():
        setattr(pkg, sub, mod)
        sys.modules[name + "." + sub] = mod
    # the names the reference's package __init__ re-exports from these modules (pytorch_pretrained_bert/__init__.py:3-6)
    for attr in ("BertConfig", "BertForPreTrainingLossMask", "BertForSeq2SeqDecoder"):
        setattr(pkg, attr, getattr(modeling, attr))
    pkg.BertAdam = optimization.BertAdam
    pkg.FP16_Optimizer_State = optimization_fp16.FP16_Optimizer_State
    sys.modules[name] = pkg
    from . import data_parallel
    misc = sys.modules.get("misc")
    if misc is None or getattr(misc, "__vlp_amd_alias__", False):
        misc = types.ModuleType("misc")
        misc.__vlp_amd_alias__ = True
        misc.__path__ = []
        sys.modules["misc"] = misc
    misc.data_parallel = data_parallel
    sys.modules["misc.data_parallel"] = data_parallel
    if with_apex_shim:
        try:
            import apex.optimizers  # noqa: F401
        except ImportError:
            apex = sys.modules.get("apex") or types.ModuleType("apex")
            apex.__path__ = []
            opt = types.ModuleType("apex.optimizers")
            opt.FusedAdam = optimization_fp16.FusedAdam
            opt.__vlp_amd_alias__ = True
            apex.optimizers = opt
            sys.modules["apex"], sys.modules["apex.optimizers"] = apex, opt
    return pkg


def uninstall():
    for k in [k for k, v in list(sys.modules.items()) if (k == "pytorch_pretrained_bert" or k.startswith("pytorch_pretrained_bert.") or
                                                           k in ("apex", "apex.optimizers", "misc", "misc.data_parallel"))
              and (getattr(v, "__vlp_amd_alias__", False) or k.startswith("pytorch_pretrained_bert.") or k == "misc.data_parallel")]:
        if k.startswith("pytorch_pretrained_bert.") and not getattr(sys.modules.get("pytorch_pretrained_bert"), "__vlp_amd_alias__", False):
            continue
        del sys.modules[k]
