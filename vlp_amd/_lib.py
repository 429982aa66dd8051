"""ctypes binding of libvlp_hip.so (include/vlp_hip.h).

The structures below mirror the header field for field.  Every wrapper takes torch tensors only to
extract raw device pointers (`data_ptr()`) and the current HIP stream; the library itself has no
torch dependency.  A failing call raises RuntimeError(vlp_last_error_string()).

The product path has NO fallback: if the shared library is missing or was not built, importing the
compute modules raises -- it never silently routes through torch ops or the CPU oracle.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLP_HIP_LIB") or os.path.join(_HERE, "libvlp_hip.so")      # VLP_HIP_LIB: A/B runs against another build of the SAME ABI

ACT_NONE, ACT_GELU, ACT_RELU, ACT_TANH, ACT_GELU_SAVE_GRAD = 0, 1, 2, 3, 4
MUL_NONE, MUL_GELU_GRAD, MUL_RELU_MASK, MUL_PLAIN = 0, 1, 2, 3

vp, i64, i32, f32, u64, u32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_uint32


class GemmNtArgs(C.Structure):
    _fields_ = [("X", vp), ("ldx", i64), ("W", vp), ("ldw", i64), ("Y", vp), ("ldy", i64), ("bias", vp),
                ("residual", vp), ("ldr", i64), ("preact", vp), ("ldp", i64), ("mul_src", vp), ("ldm", i64),
                ("M", i32), ("N", i32), ("K", i32), ("act", i32), ("mul_mode", i32), ("alpha", f32),
                ("dropout_p", f32), ("seed", u64), ("rng_stream", u32), ("variant", i32), ("row_map", vp)]


class GemmTnArgs(C.Structure):
    _fields_ = [("A", vp), ("lda", i64), ("B", vp), ("ldb", i64), ("C", vp), ("ldc", i64),
                ("M", i32), ("N", i32), ("K", i32), ("beta", i32), ("workspace", vp), ("workspace_bytes", i64),
                ("variant", i32), ("splits", i32), ("bias_out", vp)]


class ColsumArgs(C.Structure):
    _fields_ = [("A", vp), ("lda", i64), ("M", i32), ("N", i32), ("out", vp), ("beta", i32),
                ("workspace", vp), ("workspace_bytes", i64)]


class AttnFwdArgs(C.Structure):
    _fields_ = [("qkv", vp), ("ld_qkv", i64), ("mask", vp), ("ctx", vp), ("ld_ctx", i64), ("lse", vp),
                ("B", i32), ("L", i32), ("heads", i32), ("scale", f32),
                ("dropout_p", f32), ("seed", u64), ("rng_stream", u32), ("row_off", vp)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("qkv", vp), ("ld_qkv", i64), ("mask", vp), ("mask_t", vp), ("ctx", vp), ("ld_ctx", i64), ("dctx", vp), ("ld_dctx", i64),
                ("lse", vp), ("dqkv", vp), ("ld_dqkv", i64), ("delta", vp),
                ("B", i32), ("L", i32), ("heads", i32), ("scale", f32),
                ("dropout_p", f32), ("seed", u64), ("rng_stream", u32), ("row_off", vp)]


class LayerNormFwdArgs(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("gamma", vp), ("beta", vp), ("y", vp), ("ldy", i64), ("mean", vp), ("rstd", vp),
                ("M", i32), ("H", i32), ("eps", f32), ("dropout_p", f32), ("seed", u64), ("rng_stream", u32), ("row_map", vp)]


class LayerNormBwdArgs(C.Structure):
    _fields_ = [("dy", vp), ("lddy", i64), ("x", vp), ("ldx", i64), ("gamma", vp), ("mean", vp), ("rstd", vp),
                ("dx", vp), ("lddx", i64), ("dx_drop", vp), ("lddxd", i64), ("dgamma", vp), ("dbeta", vp),
                ("M", i32), ("H", i32), ("beta", i32),
                ("dy_drop_p", f32), ("dy_seed", u64), ("dy_stream", u32),
                ("out_drop_p", f32), ("out_seed", u64), ("out_stream", u32),
                ("workspace", vp), ("workspace_bytes", i64), ("defer_reduce", i32), ("row_map", vp)]


class EmbedFwdArgs(C.Structure):
    _fields_ = [("input_ids", vp), ("segment_ids", vp), ("word_emb", vp), ("pos_emb", vp), ("type_emb", vp),
                ("vis_h", vp), ("vispe_h", vp), ("pre", vp),
                ("B", i32), ("L", i32), ("Nv", i32), ("H", i32), ("vocab", i32), ("type_vocab", i32),
                ("position_ids", vp), ("max_pos", i32), ("region_mask", vp), ("row_map", vp), ("rows", i32)]


class AttnDecodeArgs(C.Structure):
    _fields_ = [("q", vp), ("ld_q", i64), ("q_rows_per_batch", i64), ("k", vp), ("v", vp), ("ld_kv", i64), ("kv_rows_per_batch", i64),
                ("mask", vp), ("ctx", vp), ("ld_ctx", i64), ("B", i32), ("Lq", i32), ("Lk", i32), ("heads", i32), ("scale", f32),
                ("k_prefix", vp), ("v_prefix", vp), ("prefix_rows_per_batch", i64), ("n_prefix", i32), ("beams", i32)]


class DecGemmArgs(C.Structure):
    _fields_ = [("X", vp), ("ldx", i64), ("W", vp), ("ldw", i64), ("bias", vp), ("Y", vp), ("ldy", i64), ("slab", vp), ("ldslab", i64),
                ("kv_cache", vp), ("kv_ld", i64), ("kv_col0", i32), ("kv_Lcap", i32), ("kv_T", i32), ("kv_start", i32),
                ("M", i32), ("N", i32), ("K", i32), ("splits", i32), ("act", i32),
                ("residual", vp), ("ldr", i64), ("ln_gamma", vp), ("ln_beta", vp), ("ln_eps", f32), ("ln_out", vp), ("ld_ln_out", i64)]


class DecReduceLnArgs(C.Structure):
    _fields_ = [("slab", vp), ("ldslab", i64), ("splits", i32), ("bias", vp), ("residual", vp), ("ldr", i64), ("gamma", vp), ("beta", vp),
                ("eps", f32), ("Y", vp), ("ldy", i64), ("M", i32), ("H", i32)]


class VisPePrepArgs(C.Structure):
    _fields_ = [("bbox", vp), ("cls", vp), ("ld_cls", i64), ("out", vp), ("ld_out", i64), ("B", i32), ("Nv", i32), ("n_cls", i32),
                ("pad_to", i32), ("cls_is_f32", i32), ("eps", f32)]


class BeamSelectArgs(C.Structure):
    _fields_ = [("kk_scores", vp), ("kk_ids", vp), ("last_total", vp), ("last_eos", vp), ("out_scores", vp), ("out_ids", vp),
                ("out_ptrs", vp), ("out_eos", vp), ("src_rows", vp), ("next_ids", vp), ("next_ids_stride", i64),
                ("B", i32), ("K", i32), ("first", i32), ("eos_id", i64)]


class EmbedBwdArgs(C.Structure):
    _fields_ = [("dpre", vp), ("input_ids", vp), ("segment_ids", vp), ("vis_h", vp), ("vispe_h", vp),
                ("d_word_emb", vp), ("d_pos_emb", vp), ("d_type_emb", vp), ("d_vis_h", vp), ("d_vispe_h", vp), ("acc32", vp),
                ("B", i32), ("L", i32), ("Nv", i32), ("H", i32), ("vocab", i32), ("type_vocab", i32),
                ("drop_p", f32), ("seed", u64), ("vis_stream", u32), ("vispe_stream", u32), ("region_mask", vp), ("parts", i32)]


class PretextFwdArgs(C.Structure):
    _fields_ = [("vis_h", vp), ("vispe_h", vp), ("pooled", vp), ("vis_masked_pos", vp), ("probs", vp), ("sample_loss", vp), ("loss", vp),
                ("B", i32), ("Nv", i32), ("Pm", i32), ("H", i32)]


class PretextBwdArgs(C.Structure):
    _fields_ = [("vis_h", vp), ("vispe_h", vp), ("pooled", vp), ("vis_masked_pos", vp), ("probs", vp), ("gscale", vp),
                ("d_vis_h", vp), ("d_vispe_h", vp), ("d_pooled_pre", vp), ("B", i32), ("Nv", i32), ("Pm", i32), ("H", i32),
                ("drop_p", f32), ("seed", u64), ("vis_stream", u32), ("vispe_stream", u32)]


class TransposeDesc(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("lds", i64), ("ldd", i64), ("rows", i32), ("cols", i32), ("rows_pad", i32), ("reserved", i32)]


class MlmLossFwdArgs(C.Structure):
    _fields_ = [("logits", vp), ("ld_logits", i64), ("labels", vp), ("weights", vp), ("loss", vp), ("lse", vp), ("coef", vp),
                ("row_loss", vp), ("B", i32), ("P", i32), ("V", i32), ("drop_worst_ratio", f32)]


class MlmLossBwdArgs(C.Structure):
    _fields_ = [("logits", vp), ("ld_logits", i64), ("labels", vp), ("lse", vp), ("coef", vp), ("grad_scale", vp),
                ("dlogits", vp), ("ld_dlogits", i64), ("rows", i32), ("V", i32)]


class FusedAdamArgs(C.Structure):
    _fields_ = [("p32", vp), ("m", vp), ("v", vp), ("g16", vp), ("p16", vp), ("n", i64),
                ("b1", f32), ("b2", f32), ("eps", f32), ("decay", f32), ("eps_inside_sqrt", i32), ("hyper", vp)]


class BertAdamArgs(C.Structure):
    _fields_ = [("p32", vp), ("m", vp), ("v", vp), ("g", vp), ("g_is_f32", i32), ("p16", vp),
                ("seg_off", vp), ("ntensors", i32), ("n", i64), ("norms", vp), ("norms_floats", i64),
                ("lr", f32), ("b1", f32), ("b2", f32), ("eps", f32), ("decay", f32), ("max_grad_norm", f32), ("grad_scale", f32),
                ("active", vp)]


# every symbol include/vlp_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "vlp_version": (C.c_int, []),
    "vlp_last_error_string": (C.c_char_p, []),
    "vlp_lab_build": (C.c_int, []),
    "vlp_debug_device_lookup_stats": (None, [vp, vp]),
    "vlp_gemm_nt": (C.c_int, [C.POINTER(GemmNtArgs), vp]),
    "vlp_gemm_nt_resolved_variant": (C.c_int, []),
    "vlp_gemm_nt_splitk_workspace_bytes": (C.c_int64, [i32, i32, i32]),
    "vlp_gemm_nt_splitk": (C.c_int, [C.POINTER(GemmNtArgs), i32, vp, i64, vp]),
    "vlp_gemm_tn_workspace_bytes": (i64, [i32, i32, i32]),
    "vlp_gemm_tn": (C.c_int, [C.POINTER(GemmTnArgs), vp]),
    "vlp_gemm_tn_grouped": (C.c_int, [C.POINTER(GemmTnArgs), i32, vp]),
    "vlp_gemm_tn_grouped_workspace_bytes": (i64, [i32]),
    "vlp_colsum_workspace_bytes": (i64, [i32, i32]),
    "vlp_colsum": (C.c_int, [C.POINTER(ColsumArgs), vp]),
    "vlp_attn_fwd": (C.c_int, [C.POINTER(AttnFwdArgs), vp]),
    "vlp_attn_bwd": (C.c_int, [C.POINTER(AttnBwdArgs), vp]),
    "vlp_attn_decode": (C.c_int, [C.POINTER(AttnDecodeArgs), vp]),
    "vlp_mask_pack_rect": (C.c_int, [vp, i64, i64, vp, i32, i32, i32, i32, vp]),
    "vlp_kv_append": (C.c_int, [vp, i64, vp, i32, i32, i32, i32, i32, vp]),
    "vlp_dec_gemm": (C.c_int, [C.POINTER(DecGemmArgs), vp]),
    "vlp_dec_reduce_ln": (C.c_int, [C.POINTER(DecReduceLnArgs), vp]),
    "vlp_argmax_rows2": (C.c_int, [vp, i64, i32, i32, vp, i64, vp, i64, vp, i64, vp]),
    "vlp_logsoftmax_topk": (C.c_int, [vp, i64, i32, i32, i32, vp, i32, i32, vp, vp, vp]),
    "vlp_beam_select": (C.c_int, [C.POINTER(BeamSelectArgs), vp]),
    "vlp_kv_gather": (C.c_int, [vp, i64, vp, i64, vp, i32, i32, i32, i32, vp]),
    "vlp_embed_bwd_workspace_floats": (C.c_int64, [i32, i32, i32, i32]),
    "vlp_mask_build": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp, i32, vp]),
    "vlp_vis_pe_prep": (C.c_int, [C.POINTER(VisPePrepArgs), vp]),
    "vlp_sample_rows": (C.c_int, [vp, i64, i32, i32, C.c_uint64, C.c_uint32, vp, i64, vp, i64, vp]),
    "vlp_argmax_rows": (C.c_int, [vp, i64, i32, i32, vp, i64, vp, i64, vp]),
    "vlp_mask_pack": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "vlp_layernorm_fwd": (C.c_int, [C.POINTER(LayerNormFwdArgs), vp]),
    "vlp_layernorm_bwd_workspace_bytes": (i64, [i32]),
    "vlp_layernorm_bwd": (C.c_int, [C.POINTER(LayerNormBwdArgs), vp]),
    "vlp_layernorm_bwd_reduce_batched": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "vlp_embed_fwd": (C.c_int, [C.POINTER(EmbedFwdArgs), vp]),
    "vlp_embed_bwd": (C.c_int, [C.POINTER(EmbedBwdArgs), vp]),
    "vlp_region_mask_build": (C.c_int, [vp, i32, i32, i32, vp, vp]),
    "vlp_pretext_fwd": (C.c_int, [C.POINTER(PretextFwdArgs), vp]),
    "vlp_pretext_bwd": (C.c_int, [C.POINTER(PretextBwdArgs), vp]),
    "vlp_copy2d": (C.c_int, [vp, i64, i32, vp, i64, i32, i32, i32, i32, vp]),
    "vlp_transpose": (C.c_int, [vp, i64, vp, i64, i32, i32, i32, vp]),
    "vlp_transpose_batched": (C.c_int, [vp, vp, i32, i32, vp]),
    "vlp_gather_rows": (C.c_int, [vp, i64, vp, vp, i64, i32, i32, i32, i32, vp, vp]),
    "vlp_scatter_add_rows": (C.c_int, [vp, i64, vp, vp, i64, i32, i32, i32, i32, vp, vp]),
    "vlp_rowmap_build": (C.c_int, [vp, i32, i32, vp, vp]),
    "vlp_rows_unpack": (C.c_int, [vp, i64, vp, i32, vp, i64, i32, vp]),
    "vlp_rows_pack": (C.c_int, [vp, i64, vp, i32, vp, i64, i32, vp]),
    "vlp_vqa_mul_fwd": (C.c_int, [vp, vp, i32, i32, i32, i32, vp, vp]),
    "vlp_vqa_mul_bwd": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "vlp_relu_dropout_bwd": (C.c_int, [vp, vp, vp, i64, i64, f32, u64, u32, vp]),
    "vlp_gelu_bwd": (C.c_int, [vp, vp, vp, i64, vp]),
    "vlp_mlm_loss_fwd": (C.c_int, [C.POINTER(MlmLossFwdArgs), vp]),
    "vlp_mlm_loss_bwd": (C.c_int, [C.POINTER(MlmLossBwdArgs), vp]),
    "vlp_bce_loss_fwd": (C.c_int, [vp, i64, vp, i64, i32, i32, vp, vp]),
    "vlp_bce_loss_bwd": (C.c_int, [vp, i64, vp, i64, i32, i32, vp, vp, i64, vp]),
    "vlp_sumsq": (C.c_int, [vp, i64, vp, vp, vp]),
    "vlp_sumsq_acc": (C.c_int, [vp, i64, vp, vp, vp]),
    "vlp_fused_adam": (C.c_int, [C.POINTER(FusedAdamArgs), vp]),
    "vlp_adam_hyper": (C.c_int, [vp, vp, vp, f32, f32, vp, vp]),
    "vlp_loss_scale_update": (C.c_int, [vp, vp, vp]),
    "vlp_bert_adam": (C.c_int, [C.POINTER(BertAdamArgs), vp]),
    "vlp_bert_adam_norms_floats": (i64, [i64, i32]),
}

_lib = None


# ---- launch plans ----------------------------------------------------------------------------------------------------
# The incremental decoder issues ~160 tiny launches per token step whose arguments (workspace pointers, shapes) are identical from
# call to call; building the ctypes structs and looking up strides every time made it host-bound (7 us per launch).  record() collects
# the C calls a block of Python makes, replay() re-issues them: ~1 us per launch.  A plan holds its argument objects alive; it is only
# valid while every buffer it points to is (the engine keeps plans inside the workspace they were recorded against).
_recording = None


class _Recorder(object):
    """Stands in for the loaded library while recording: every C call is executed AND appended to the plan."""

    def __init__(self, lib, plan):
        self._lib, self._plan = lib, plan

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*args):
            self._plan.append((fn, args))
            return fn(*args)
        return call


class record(object):
    """with _lib.record() as plan: ...   (plan = list of (cfunc, args))"""

    def __enter__(self):
        global _recording
        if _recording is not None:
            raise RuntimeError("vlp_amd._lib.record(): already recording")
        self.plan = []
        _recording = _Recorder(load(), self.plan)
        return self.plan

    def __exit__(self, *exc):
        global _recording
        _recording = None
        return False


def replay(plan):
    for fn, args in plan:
        rc = fn(*args)
        if rc:
            _check(rc)


def load():
    """Load libvlp_hip.so (built by vlp_amd/build.py).  Raises if it is absent: there is no fallback."""
    global _lib
    if _recording is not None:
        return _recording
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libvlp_hip.so not found at %s -- run `python -m vlp_amd.build` (or "
                           "__graft_entry__.build()).  vlp_amd has no CPU / torch fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.vlp_version() != 5:      # include/vlp_hip.h VLP_ABI_VERSION
        raise RuntimeError("libvlp_hip.so ABI version mismatch")
    _lib = lib
    return lib


def lab_build():
    """True when the loaded library carries the investigation variants (-DVLP_LAB_BUILD)."""
    return bool(load().vlp_lab_build())


def _check(rc):
    if rc != 0:
        lib = _lib if _lib is not None else load()
        raise RuntimeError("libvlp_hip: %s (status %d)" % (lib.vlp_last_error_string().decode(), rc))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vlp_amd kernels need device tensors (got a CPU tensor); there is no CPU fallback")


# --------------------------------------------------------------------------------------------------
# thin typed wrappers
# --------------------------------------------------------------------------------------------------
def gemm_nt(x, w, y, M, N, K, ldx=None, ldw=None, ldy=None, bias=None, residual=None, ldr=None, preact=None, ldp=None,
            mul_src=None, ldm=None, act=ACT_NONE, mul_mode=MUL_NONE, alpha=1.0, dropout_p=0.0, seed=0, rng_stream=0,
            variant=0, row_map=None):
    _req_cuda(x, w, y, row_map)
    a = GemmNtArgs(ptr(x), ldx if ldx is not None else x.stride(0), ptr(w), ldw if ldw is not None else w.stride(0),
                   ptr(y), ldy if ldy is not None else y.stride(0), ptr(bias),
                   ptr(residual), (ldr if ldr is not None else (residual.stride(0) if residual is not None else 0)),
                   ptr(preact), (ldp if ldp is not None else (preact.stride(0) if preact is not None else 0)),
                   ptr(mul_src), (ldm if ldm is not None else (mul_src.stride(0) if mul_src is not None else 0)),
                   M, N, K, act, mul_mode, alpha, dropout_p, seed, rng_stream, variant, ptr(row_map))
    _check(load().vlp_gemm_nt(C.byref(a), stream_ptr()))


def gemm_nt_resolved_variant():
    """The variant this thread's last gemm_nt actually launched (after the launcher's fallbacks)."""
    return int(load().vlp_gemm_nt_resolved_variant())


def gemm_nt_splitk_workspace_bytes(M, N, splits):
    return int(load().vlp_gemm_nt_splitk_workspace_bytes(M, N, splits))


def gemm_nt_splitk(x, w, y, M, N, K, splits, workspace, ldx=None, ldw=None, ldy=None, bias=None, residual=None, ldr=None, preact=None, ldp=None,
                   mul_src=None, ldm=None, act=ACT_NONE, mul_mode=MUL_NONE, alpha=1.0, dropout_p=0.0, seed=0, rng_stream=0):
    _req_cuda(x, w, y, workspace)
    a = GemmNtArgs(ptr(x), ldx if ldx is not None else x.stride(0), ptr(w), ldw if ldw is not None else w.stride(0),
                   ptr(y), ldy if ldy is not None else y.stride(0), ptr(bias),
                   ptr(residual), (ldr if ldr is not None else (residual.stride(0) if residual is not None else 0)),
                   ptr(preact), (ldp if ldp is not None else (preact.stride(0) if preact is not None else 0)),
                   ptr(mul_src), (ldm if ldm is not None else (mul_src.stride(0) if mul_src is not None else 0)),
                   M, N, K, act, mul_mode, alpha, dropout_p, seed, rng_stream, 0, None)
    _check(load().vlp_gemm_nt_splitk(C.byref(a), splits, ptr(workspace), workspace.numel() * workspace.element_size(), stream_ptr()))


def gemm_tn_workspace_bytes(M, N, K):
    return int(load().vlp_gemm_tn_workspace_bytes(M, N, K))


def gemm_tn(a_, b_, c_, M, N, K, lda=None, ldb=None, ldc=None, beta=0, workspace=None, variant=0, splits=0, bias_out=None):
    _req_cuda(a_, b_, c_)
    a = GemmTnArgs(ptr(a_), lda if lda is not None else a_.stride(0), ptr(b_), ldb if ldb is not None else b_.stride(0),
                   ptr(c_), ldc if ldc is not None else c_.stride(0), M, N, K, beta,
                   ptr(workspace), workspace.numel() * workspace.element_size() if workspace is not None else 0, variant, splits,
                   ptr(bias_out))
    _check(load().vlp_gemm_tn(C.byref(a), stream_ptr()))


def gemm_tn_grouped_workspace_bytes(tiles):
    return int(load().vlp_gemm_tn_grouped_workspace_bytes(tiles))


def gemm_tn_grouped(problems, workspace=None):
    """problems: list of tuples (a, b, c, M, N, K, beta, bias_out) -- several wgrads in one launch (vlp_gemm_tn_grouped); `workspace`
    (gemm_tn_grouped_workspace_bytes) lets the launcher use its stream-K form."""
    arr = (GemmTnArgs * len(problems))()
    for i, (a_, b_, c_, M, N, K, beta, bias_out) in enumerate(problems):
        _req_cuda(a_, b_, c_, bias_out)
        w = workspace if i == 0 else None
        if w is not None:
            _req_cuda(w)
        arr[i] = GemmTnArgs(ptr(a_), a_.stride(0), ptr(b_), b_.stride(0), ptr(c_), c_.stride(0), M, N, K, beta, ptr(w),
                            (w.numel() * w.element_size()) if w is not None else 0, 0, 0, ptr(bias_out))
    _check(load().vlp_gemm_tn_grouped(arr, len(problems), stream_ptr()))


def colsum_workspace_bytes(M, N):
    return int(load().vlp_colsum_workspace_bytes(M, N))


def colsum(a_, out, M, N, lda=None, beta=0, workspace=None):
    _req_cuda(a_, out)
    a = ColsumArgs(ptr(a_), lda if lda is not None else a_.stride(0), M, N, ptr(out), beta, ptr(workspace),
                   workspace.numel() * workspace.element_size())
    _check(load().vlp_colsum(C.byref(a), stream_ptr()))


def attn_fwd(qkv, mask, ctx, lse, B, L, heads, scale, dropout_p=0.0, seed=0, rng_stream=0, row_off=None):
    """row_off (int32 [B+1], device): packed rows -- sample b owns rows [row_off[b], row_off[b+1]) of qkv / ctx (include/vlp_hip.h)."""
    _req_cuda(qkv, mask, ctx, lse, row_off)
    a = AttnFwdArgs(ptr(qkv), qkv.stride(0), ptr(mask), ptr(ctx), ctx.stride(0), ptr(lse), B, L, heads, scale, dropout_p, seed, rng_stream,
                    ptr(row_off))
    _check(load().vlp_attn_fwd(C.byref(a), stream_ptr()))


def attn_bwd(qkv, mask, mask_t, ctx, dctx, lse, dqkv, delta, B, L, heads, scale, dropout_p=0.0, seed=0, rng_stream=0, row_off=None):
    _req_cuda(qkv, mask, mask_t, ctx, dctx, lse, dqkv, delta, row_off)
    a = AttnBwdArgs(ptr(qkv), qkv.stride(0), ptr(mask), ptr(mask_t), ptr(ctx), ctx.stride(0), ptr(dctx), dctx.stride(0), ptr(lse),
                    ptr(dqkv), dqkv.stride(0), ptr(delta), B, L, heads, scale, dropout_p, seed, rng_stream, ptr(row_off))
    _check(load().vlp_attn_bwd(C.byref(a), stream_ptr()))


def attn_decode(q, ld_q, q_rows, k, v, ld_kv, kv_rows, mask, ctx, B, Lq, Lk, heads, scale, k_prefix=None, v_prefix=None, prefix_rows=0,
                n_prefix=0, beams=1):
    _req_cuda(q, k, v, mask, ctx, k_prefix, v_prefix)
    a = AttnDecodeArgs(ptr(q), ld_q, q_rows, ptr(k), ptr(v), ld_kv, kv_rows, ptr(mask), ptr(ctx), ctx.stride(0), B, Lq, Lk, heads, scale,
                       ptr(k_prefix), ptr(v_prefix), prefix_rows, n_prefix, beams)
    _check(load().vlp_attn_decode(C.byref(a), stream_ptr()))


def dec_gemm(x, w, M, N, Kd, y=None, bias=None, act=0, slab=None, splits=1, kv_cache=None, kv_col0=0, kv_Lcap=0, kv_T=1, kv_start=0,
             residual=None, ln_gamma=None, ln_beta=None, ln_eps=1e-5, ln_out=None):
    """Token-step Linear (csrc/decode.hip): y [M, N] fp16 (optionally K | V columns >= kv_col0 into kv_cache [seq, kv_Lcap, ld]) or, with
    `slab` (fp32 [splits, M, ldslab]), raw split-K partial sums for dec_reduce_ln."""
    _req_cuda(x, w, y, bias, slab, kv_cache, residual, ln_gamma, ln_beta, ln_out)
    a = DecGemmArgs(ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(y), y.stride(0) if y is not None else 0,
                    ptr(slab), slab.stride(1) if slab is not None else 0, ptr(kv_cache), kv_cache.stride(1) if kv_cache is not None else 0,
                    kv_col0, kv_Lcap, kv_T, kv_start, M, N, Kd, splits, act,
                    ptr(residual), residual.stride(0) if residual is not None else 0, ptr(ln_gamma), ptr(ln_beta), ln_eps,
                    ptr(ln_out), ln_out.stride(0) if ln_out is not None else 0)
    _check(load().vlp_dec_gemm(C.byref(a), stream_ptr()))


def dec_reduce_ln(slab, splits, bias, residual, gamma, beta, y, M, H, eps=1e-5):
    _req_cuda(slab, bias, residual, gamma, beta, y)
    a = DecReduceLnArgs(ptr(slab), slab.stride(1), splits, ptr(bias), ptr(residual), residual.stride(0) if residual is not None else 0,
                        ptr(gamma), ptr(beta), eps, ptr(y), y.stride(0), M, H)
    _check(load().vlp_dec_reduce_ln(C.byref(a), stream_ptr()))


def argmax_rows2(logits, ld, rows, V, ids_a, ids_b, vals):
    _req_cuda(logits, ids_a, ids_b, vals)
    _check(load().vlp_argmax_rows2(ptr(logits), ld, rows, V, ptr(ids_a), ids_a.stride(0), ptr(ids_b), ids_b.stride(0) if ids_b is not None else 0,
                                   ptr(vals), vals.stride(0), stream_ptr()))


def mask_pack_rect(mask_view, out_u8, B, Lq, Lk, Lkp):
    """mask_view: int64 tensor view [B, Lq, Lk] with unit column stride (any batch / row strides)."""
    _req_cuda(mask_view, out_u8)
    assert mask_view.stride(2) == 1 and mask_view.dtype == torch.int64
    _check(load().vlp_mask_pack_rect(ptr(mask_view), mask_view.stride(0), mask_view.stride(1), ptr(out_u8), B, Lq, Lk, Lkp, stream_ptr()))


def kv_append(qkv_new, ld, cache, Lcap, B, T, start, H):
    _req_cuda(qkv_new, cache)
    _check(load().vlp_kv_append(ptr(qkv_new), ld, ptr(cache), Lcap, B, T, start, H, stream_ptr()))


def argmax_rows(logits, ld, rows, V, ids, vals):
    """ids / vals: 1-D (possibly strided) views with `rows` elements."""
    _req_cuda(logits, ids, vals)
    _check(load().vlp_argmax_rows(ptr(logits), ld, rows, V, ptr(ids), ids.stride(0), ptr(vals), vals.stride(0), stream_ptr()))


def logsoftmax_topk(logits, ld, rows, V, K, out_scores, out_ids, forbid=None, eos_id=0, block_eos=False):
    _req_cuda(logits, out_scores, out_ids, forbid)
    _check(load().vlp_logsoftmax_topk(ptr(logits), ld, rows, V, K, ptr(forbid), eos_id, 1 if block_eos else 0, ptr(out_scores), ptr(out_ids),
                                      stream_ptr()))


def beam_select(kk_scores, kk_ids, last_total, last_eos, out_scores, out_ids, out_ptrs, out_eos, src_rows, next_ids, B, K, first, eos_id):
    """next_ids: 1-D (possibly strided) int64 view with B*K elements."""
    _req_cuda(kk_scores, kk_ids, last_total, last_eos, out_scores, out_ids, out_ptrs, out_eos, src_rows, next_ids)
    a = BeamSelectArgs(ptr(kk_scores), ptr(kk_ids), ptr(last_total), ptr(last_eos), ptr(out_scores), ptr(out_ids), ptr(out_ptrs), ptr(out_eos),
                       ptr(src_rows), ptr(next_ids), next_ids.stride(0), B, K, 1 if first else 0, eos_id)
    _check(load().vlp_beam_select(C.byref(a), stream_ptr()))


def kv_gather(src, src_rows, dst, dst_rows, idx, R, lo, hi, row_elems):
    _req_cuda(src, dst, idx)
    _check(load().vlp_kv_gather(ptr(src), src_rows, ptr(dst), dst_rows, ptr(idx), R, lo, hi, row_elems, stream_ptr()))


def mask_build(second_st, second_end, is_s2s, out_u8, B, L, Lp, out_t=None, region_mask=None, Nv=0):
    """second_st / second_end / is_s2s: int32 [B] device tensors; region_mask: u8 [B*Nv] (region_mask_build) -> those key columns are blocked."""
    _req_cuda(second_st, second_end, is_s2s, out_u8, out_t, region_mask)
    for t in (second_st, second_end, is_s2s):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.numel() == B
    _check(load().vlp_mask_build(ptr(second_st), ptr(second_end), ptr(is_s2s), ptr(out_u8), ptr(out_t), B, L, Lp, ptr(region_mask), Nv, stream_ptr()))


def vis_pe_prep(bbox, cls, out, B, Nv, n_cls, pad_to, eps=1e-5):
    """bbox f32 [B,Nv,6]; cls f16/f32 [B*Nv, >=n_cls] (row stride = stride(0)); out f16 [B*Nv, >=pad_to]."""
    _req_cuda(bbox, cls, out)
    assert bbox.dtype == torch.float32 and bbox.is_contiguous() and cls.dtype in (torch.float16, torch.float32) and cls.stride(-1) == 1
    a = VisPePrepArgs(ptr(bbox), ptr(cls), cls.stride(0), ptr(out), out.stride(0), B, Nv, n_cls, pad_to, 1 if cls.dtype == torch.float32 else 0, eps)
    _check(load().vlp_vis_pe_prep(C.byref(a), stream_ptr()))


def sample_rows(logits, ld, rows, V, seed, rng_stream, ids, logp):
    """ids / logp: 1-D (possibly strided) views with `rows` elements."""
    _req_cuda(logits, ids, logp)
    _check(load().vlp_sample_rows(ptr(logits), ld, rows, V, seed, rng_stream, ptr(ids), ids.stride(0), ptr(logp), logp.stride(0), stream_ptr()))


def mask_pack(mask_i64, out_u8, B, L, Lp, out_t=None):
    _req_cuda(mask_i64, out_u8, out_t)
    _check(load().vlp_mask_pack(ptr(mask_i64), ptr(out_u8), ptr(out_t), B, L, Lp, stream_ptr()))


def layernorm_fwd(x, gamma, beta, y, M, H, mean=None, rstd=None, eps=1e-5, dropout_p=0.0, seed=0, rng_stream=0, ldx=None, ldy=None, row_map=None):
    _req_cuda(x, gamma, beta, y, row_map)
    a = LayerNormFwdArgs(ptr(x), ldx if ldx is not None else x.stride(0), ptr(gamma), ptr(beta), ptr(y),
                         ldy if ldy is not None else y.stride(0), ptr(mean), ptr(rstd), M, H, eps, dropout_p, seed, rng_stream, ptr(row_map))
    _check(load().vlp_layernorm_fwd(C.byref(a), stream_ptr()))


def layernorm_bwd_workspace_bytes(H):
    return int(load().vlp_layernorm_bwd_workspace_bytes(H))


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, M, H, workspace, beta=0, dx_drop=None,
                  dy_drop=(0.0, 0, 0), out_drop=(0.0, 0, 0), defer_reduce=False, row_map=None):
    """defer_reduce: leave the dgamma / dbeta partials in `workspace` (one private slot per LayerNorm) for layernorm_bwd_reduce_batched."""
    _req_cuda(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, row_map)
    a = LayerNormBwdArgs(ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), dx.stride(0),
                         ptr(dx_drop), dx_drop.stride(0) if dx_drop is not None else 0, ptr(dgamma), ptr(dbeta), M, H, beta,
                         dy_drop[0], dy_drop[1], dy_drop[2], out_drop[0], out_drop[1], out_drop[2],
                         ptr(workspace), workspace.numel() * workspace.element_size(), 1 if defer_reduce else 0, ptr(row_map))
    _check(load().vlp_layernorm_bwd(C.byref(a), stream_ptr()))


def layernorm_bwd_reduce_batched(parts, dst_table, count, M, H, beta=0):
    """parts: uint8/float32 buffer of `count` slots of layernorm_bwd_workspace_bytes(H); dst_table: int64 device tensor [count, 2] of
    dgamma / dbeta addresses."""
    _req_cuda(parts, dst_table)
    _check(load().vlp_layernorm_bwd_reduce_batched(ptr(parts), ptr(dst_table), count, M, H, beta, stream_ptr()))


def embed_fwd(input_ids, segment_ids, word_emb, pos_emb, type_emb, vis_h, vispe_h, pre, B, L, Nv, H, position_ids=None, region_mask=None,
              row_map=None, rows=0):
    """row_map (int32 [rows], device): packed output -- row p of `pre` is logical row row_map[p] = b*L + l."""
    _req_cuda(input_ids, segment_ids, word_emb, pos_emb, type_emb, pre, position_ids, region_mask, row_map)
    a = EmbedFwdArgs(ptr(input_ids), ptr(segment_ids), ptr(word_emb), ptr(pos_emb), ptr(type_emb), ptr(vis_h), ptr(vispe_h),
                     ptr(pre), B, L, Nv, H, word_emb.shape[0], type_emb.shape[0], ptr(position_ids), pos_emb.shape[0], ptr(region_mask),
                     ptr(row_map), rows)
    _check(load().vlp_embed_fwd(C.byref(a), stream_ptr()))


def embed_bwd_workspace_floats(B, L, Nv, H):
    return int(load().vlp_embed_bwd_workspace_floats(B, L, Nv, H))


def embed_bwd(dpre, input_ids, segment_ids, vis_h, vispe_h, d_word, d_pos, d_type, d_vis_h, d_vispe_h, acc32,
              B, L, Nv, H, vocab, type_vocab, drop_p=0.0, seed=0, vis_stream=0, vispe_stream=0, region_mask=None, parts=0):
    """parts: 0 = everything, 1 = region rows only (d_vis_h / d_vispe_h), 2 = the embedding tables only."""
    _req_cuda(dpre, input_ids, segment_ids, d_word, d_pos, d_type, acc32, region_mask)
    a = EmbedBwdArgs(ptr(dpre), ptr(input_ids), ptr(segment_ids), ptr(vis_h), ptr(vispe_h), ptr(d_word), ptr(d_pos), ptr(d_type),
                     ptr(d_vis_h), ptr(d_vispe_h), ptr(acc32), B, L, Nv, H, vocab, type_vocab, drop_p, seed, vis_stream, vispe_stream,
                     ptr(region_mask), parts)
    _check(load().vlp_embed_bwd(C.byref(a), stream_ptr()))


def region_mask_build(vis_masked_pos, out, B, Pm, Nv):
    """vis_masked_pos i64 [B, Pm] (1..Nv) -> out u8 [B*Nv], 1 on masked region rows (seq2seq_loader.py:267-269)."""
    _req_cuda(vis_masked_pos, out)
    _check(load().vlp_region_mask_build(ptr(vis_masked_pos), B, Pm, Nv, ptr(out), stream_ptr()))


def pretext_fwd(vis_h, vispe_h, pooled, vis_masked_pos, probs, sample_loss, loss, B, Nv, Pm, H):
    _req_cuda(vis_h, vispe_h, pooled, vis_masked_pos, probs, sample_loss, loss)
    a = PretextFwdArgs(ptr(vis_h), ptr(vispe_h), ptr(pooled), ptr(vis_masked_pos), ptr(probs), ptr(sample_loss), ptr(loss), B, Nv, Pm, H)
    _check(load().vlp_pretext_fwd(C.byref(a), stream_ptr()))


def pretext_bwd(vis_h, vispe_h, pooled, vis_masked_pos, probs, gscale, d_vis_h, d_vispe_h, d_pooled_pre, B, Nv, Pm, H,
                drop_p=0.0, seed=0, vis_stream=0, vispe_stream=0):
    _req_cuda(vis_h, vispe_h, pooled, vis_masked_pos, probs, gscale, d_vis_h, d_vispe_h, d_pooled_pre)
    a = PretextBwdArgs(ptr(vis_h), ptr(vispe_h), ptr(pooled), ptr(vis_masked_pos), ptr(probs), ptr(gscale), ptr(d_vis_h), ptr(d_vispe_h),
                       ptr(d_pooled_pre), B, Nv, Pm, H, drop_p, seed, vis_stream, vispe_stream)
    _check(load().vlp_pretext_bwd(C.byref(a), stream_ptr()))


def copy2d(src, lds, src_f32, dst, ldd, rows, cols_src, cols_dst, beta=0):
    _req_cuda(src, dst)
    _check(load().vlp_copy2d(ptr(src), lds, int(src_f32), ptr(dst), ldd, rows, cols_src, cols_dst, beta, stream_ptr()))


def transpose(src, lds, dst, ldd, rows, cols, rows_pad):
    _req_cuda(src, dst)
    _check(load().vlp_transpose(ptr(src), lds, ptr(dst), ldd, rows, cols, rows_pad, stream_ptr()))


def make_transpose_batch(items, device):
    """items: [(src, lds, dst, ldd, rows, cols, rows_pad)] -> (descs_dev, tile_start_dev, n, total_tiles).  The descriptor
    table is built once (pointers of the flat parameter buffers and shadows are stable) and reused every step."""
    import numpy as np
    arr = (TransposeDesc * len(items))()
    starts, tot = [], 0
    for i, (src, lds, dst, ldd, rows, cols, rows_pad) in enumerate(items):
        _req_cuda(src, dst)
        arr[i] = TransposeDesc(src.data_ptr(), dst.data_ptr(), lds, ldd, rows, cols, rows_pad, 0)
        starts.append(tot)
        tot += ((rows_pad + 63) // 64) * ((cols + 63) // 64)
    raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
    descs = torch.from_numpy(raw).to(device)
    ts = torch.tensor(starts, dtype=torch.int32, device=device)
    return descs, ts, len(items), tot


def transpose_batched(batch):
    descs, ts, n, tot = batch
    _check(load().vlp_transpose_batched(ptr(descs), ptr(ts), n, tot, stream_ptr()))


def gather_rows(src, lds, pos, out, ldo, B, P, L, H, row_off=None):
    _req_cuda(src, pos, out, row_off)
    _check(load().vlp_gather_rows(ptr(src), lds, ptr(pos), ptr(out), ldo, B, P, L, H, ptr(row_off), stream_ptr()))


def scatter_add_rows(src, lds, pos, dst, ldd, B, P, L, H, row_off=None):
    _req_cuda(src, pos, dst, row_off)
    _check(load().vlp_scatter_add_rows(ptr(src), lds, ptr(pos), ptr(dst), ldd, B, P, L, H, ptr(row_off), stream_ptr()))


def rowmap_build(row_off, B, L, row_map):
    """row_map[row_off[b] + l] = b*L + l for l < row_off[b+1] - row_off[b]  (packed rows -> logical rows)."""
    _req_cuda(row_off, row_map)
    _check(load().vlp_rowmap_build(ptr(row_off), B, L, ptr(row_map), stream_ptr()))


def rows_unpack(src, row_map, rows, dst, H):
    """dst[row_map[p]] = src[p]; untouched dst rows keep their contents (clear dst first)."""
    _req_cuda(src, row_map, dst)
    _check(load().vlp_rows_unpack(ptr(src), src.stride(0), ptr(row_map), rows, ptr(dst), dst.stride(0), H, stream_ptr()))


def rows_pack(src, row_map, rows, dst, H):
    """dst[p] = src[row_map[p]]"""
    _req_cuda(src, row_map, dst)
    _check(load().vlp_rows_pack(ptr(src), src.stride(0), ptr(row_map), rows, ptr(dst), dst.stride(0), H, stream_ptr()))


def vqa_mul_fwd(h, out, B, L, Nv, H, row_off=None):
    _req_cuda(h, out, row_off)
    _check(load().vlp_vqa_mul_fwd(ptr(h), ptr(out), B, L, Nv, H, ptr(row_off), stream_ptr()))


def vqa_mul_bwd(h, dout, dh, B, L, Nv, H, row_off=None):
    _req_cuda(h, dout, dh, row_off)
    _check(load().vlp_vqa_mul_bwd(ptr(h), ptr(dout), ptr(dh), B, L, Nv, H, ptr(row_off), stream_ptr()))


def relu_dropout_bwd(dy, y, dz, n, ncols, drop_p=0.0, seed=0, rng_stream=0):
    _req_cuda(dy, y, dz)
    _check(load().vlp_relu_dropout_bwd(ptr(dy), ptr(y), ptr(dz), n, ncols, drop_p, seed, rng_stream, stream_ptr()))


def gelu_bwd(dy, z, dz, n):
    _req_cuda(dy, z, dz)
    _check(load().vlp_gelu_bwd(ptr(dy), ptr(z), ptr(dz), n, stream_ptr()))


def mlm_loss_fwd(logits, ld, labels, weights, loss, lse, coef, row_loss, B, P, V, drop_worst_ratio=0.0):
    _req_cuda(logits, labels, weights, loss, lse, coef, row_loss)
    a = MlmLossFwdArgs(ptr(logits), ld, ptr(labels), ptr(weights), ptr(loss), ptr(lse), ptr(coef), ptr(row_loss), B, P, V, drop_worst_ratio)
    _check(load().vlp_mlm_loss_fwd(C.byref(a), stream_ptr()))


def mlm_loss_bwd(logits, ld, labels, lse, coef, grad_scale, dlogits, ldd, rows, V):
    _req_cuda(logits, labels, lse, coef, grad_scale, dlogits)
    a = MlmLossBwdArgs(ptr(logits), ld, ptr(labels), ptr(lse), ptr(coef), ptr(grad_scale), ptr(dlogits), ldd, rows, V)
    _check(load().vlp_mlm_loss_bwd(C.byref(a), stream_ptr()))


def bce_loss_fwd(logits, ld, labels, ldl, B, N, loss257):
    _req_cuda(logits, labels, loss257)
    _check(load().vlp_bce_loss_fwd(ptr(logits), ld, ptr(labels), ldl, B, N, ptr(loss257), stream_ptr()))


def bce_loss_bwd(logits, ld, labels, ldl, B, N, grad_scale, dlogits, ldd):
    _req_cuda(logits, labels, grad_scale, dlogits)
    _check(load().vlp_bce_loss_bwd(ptr(logits), ld, ptr(labels), ldl, B, N, ptr(grad_scale), ptr(dlogits), ldd, stream_ptr()))


def sumsq(g16, n, out2, partial, accumulate=False):
    _req_cuda(g16, out2, partial)
    fn = load().vlp_sumsq_acc if accumulate else load().vlp_sumsq
    _check(fn(ptr(g16), n, ptr(out2), ptr(partial), stream_ptr()))


def adam_hyper(sumsq2, any_overflow, scale_state, max_grad_norm, step_size, hyper3):
    _req_cuda(sumsq2, hyper3, scale_state)
    _check(load().vlp_adam_hyper(ptr(sumsq2), ptr(any_overflow), ptr(scale_state), max_grad_norm, step_size, ptr(hyper3), stream_ptr()))


def loss_scale_update(scale_state, overflow):
    _req_cuda(scale_state, overflow)
    _check(load().vlp_loss_scale_update(ptr(scale_state), ptr(overflow), stream_ptr()))


def fused_adam(p32, m, v, g16, p16, n, hyper, b1=0.9, b2=0.999, eps=1e-8, decay=0.0, eps_inside_sqrt=False):
    _req_cuda(p32, m, v, g16, p16, hyper)
    a = FusedAdamArgs(ptr(p32), ptr(m), ptr(v), ptr(g16), ptr(p16), n, b1, b2, eps, decay, int(eps_inside_sqrt), ptr(hyper))
    _check(load().vlp_fused_adam(C.byref(a), stream_ptr()))


def bert_adam_norms_floats(n, ntensors):
    return int(load().vlp_bert_adam_norms_floats(n, ntensors))


def bert_adam(p32, m, v, g, g_is_f32, p16, seg_off, ntensors, n, norms, lr, b1=0.9, b2=0.999, eps=1e-6, decay=0.01,
              max_grad_norm=1.0, grad_scale=1.0, active=None):
    _req_cuda(p32, m, v, g, seg_off, norms)
    a = BertAdamArgs(ptr(p32), ptr(m), ptr(v), ptr(g), int(g_is_f32), ptr(p16), ptr(seg_off), ntensors, n, ptr(norms), norms.numel(),   # the library checks the scratch size (ABI 3)
                     lr, b1, b2, eps, decay, max_grad_norm, grad_scale, ptr(active))
    _check(load().vlp_bert_adam(C.byref(a), stream_ptr()))
