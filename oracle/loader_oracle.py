"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the input preparation the reference does per sample on the CPU
(vlp/seq2seq_loader.py, Preprocess4Seq2seq.__call__), for SURVEY.md section 8(f) row N2 (on-device input prep).

Pinned against the UNMODIFIED reference pipeline run in the build container with an in-memory h5py
(oracle/ref_loader.load_reference_loader, tests/test_oracle_vs_reference.py::test_loader_oracle_vs_reference) and through the
fixtures tests/golden/loader_*.npz generated from it (oracle/make_golden.py).
"""
import numpy as np


def layer_norm_np(x, eps=1e-5):
    """torch.nn.functional.layer_norm over the last axis without affine (biased variance, eps inside the sqrt)."""
    x = np.asarray(x, dtype=np.float64)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps)


def vis_pe_prepare(bbox, cls_prob):
    """seq2seq_loader.py:338-351 for one image.  bbox [Nv, 6] = (x1, y1, x2, y2, <replaced by the relative area>, confidence),
    cls_prob [Nv, 1601].  Box corners are divided by the largest x / y seen in the image (+1e-5), column 4 becomes the clamped
    relative area, then the 6 box numbers and the 1601 class probabilities are layer-normed separately and concatenated."""
    b = np.array(bbox, dtype=np.float64)
    w_est = b[:, [0, 2]].max() + 1e-5
    h_est = b[:, [1, 3]].max() + 1e-5
    b[:, [0, 2]] /= w_est
    b[:, [1, 3]] /= h_est
    area = np.clip((b[:, 3] - b[:, 1]) * (b[:, 2] - b[:, 0]), 0, None)
    six = np.concatenate((b[:, :4], area[:, None], b[:, 5:]), axis=-1)
    return np.concatenate((layer_norm_np(six), layer_norm_np(cls_prob)), axis=-1)


def attention_mask(len_a, len_b, max_len, mode):
    """seq2seq_loader.py:292-301.  len_a = number of region placeholders, len_b = caption tokens (without the final [SEP]).
    s2s: every row sees [CLS] regions [SEP]; target rows additionally see the lower triangle of the target block (incl. the
    final [SEP]).  bi: every row sees every non-pad column."""
    st, en = len_a + 2, len_a + len_b + 3
    m = np.zeros((max_len, max_len), dtype=np.int64)
    if mode == "s2s":
        m[:, :st] = 1
        m[st:en, st:en] = np.tril(np.ones((en - st, en - st), dtype=np.int64))
    else:
        m[:, :en] = 1
    return m


def segment_ids(len_a, len_b, max_len, mode, new_segment_ids=True):
    """seq2seq_loader.py:240-246,289."""
    a, b = ((4, 5) if mode == "s2s" else (0, 1)) if new_segment_ids else (0, 1)
    s = [a] * (len_a + 2) + [b] * (len_b + 1)
    return np.asarray(s + [0] * (max_len - len(s)), dtype=np.int64)
