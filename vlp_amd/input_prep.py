"""On-device input preparation.

The reference prepares every sample on the CPU inside the DataLoader workers (vlp/seq2seq_loader.py:229-359): it converts the fp16
region features to fp32, normalises the boxes / layer-norms the 1601 class probabilities into the 1607-d `vis_pe`, and materialises an
int64 [L, L] attention mask -- 1.7 MB of fp32 + 223 KB of mask per sample that then cross PCIe and are cast back to fp16
(run_img2txt_dist.py:464-468).  Here the loader only has to deliver what is on disk plus a few integers:

    img       f16 [B, 100, 2048]                      (as stored; BertForPreTrainingLossMask already accepts fp16 features)
    RawRegions(bbox f32 [B, 100, 6], cls_prob f16 [B, 100, 1601])     in place of `vis_pe`
    MaskSpec(second_st, second_end, is_s2s)  int32 [B]                  in place of `attention_mask`

and the HIP engine builds the packed masks (vlp_mask_build) and the K-padded box/class encoding (vlp_vis_pe_prep) directly in the
buffers its kernels read.  Both objects are accepted wherever the reference API takes `vis_pe` / `attention_mask`
(BertForPreTrainingLossMask.forward); the dense tensors keep working unchanged.
"""
import collections

import torch

N_CLS = 1601            # Visual Genome object classes + background, hard-coded in the reference (seq2seq_loader.py:351)


class RawRegions(collections.namedtuple("RawRegions", ["bbox", "cls_prob"])):
    """bbox: f32 [B, Nv, 6] = (x1, y1, x2, y2, <ignored>, confidence) as read from the bbox h5 file (seq2seq_loader.py:330);
    cls_prob: f16 or f32 [B, Nv, 1601] as read from the `_cls` h5 file (:329)."""
    __slots__ = ()

    def to(self, device, non_blocking=False):
        return RawRegions(self.bbox.to(device, non_blocking=non_blocking), self.cls_prob.to(device, non_blocking=non_blocking))

    @property
    def shape(self):      # what the dense vis_pe would be
        return (self.bbox.shape[0], self.bbox.shape[1], 6 + self.cls_prob.shape[2])

    def check(self, B, Nv):
        if tuple(self.bbox.shape) != (B, Nv, 6) or self.bbox.dtype != torch.float32:
            raise RuntimeError("RawRegions.bbox must be f32 [%d, %d, 6]" % (B, Nv))
        if tuple(self.cls_prob.shape) != (B, Nv, N_CLS) or self.cls_prob.dtype not in (torch.float16, torch.float32):
            raise RuntimeError("RawRegions.cls_prob must be f16/f32 [%d, %d, %d]" % (B, Nv, N_CLS))
        if not (self.bbox.is_cuda and self.cls_prob.is_cuda):
            raise RuntimeError("RawRegions must live on the GPU (there is no CPU path)")


_MaskSpecBase = collections.namedtuple("MaskSpec", ["second_st", "second_end", "is_s2s", "lens_host"])
_MaskSpecBase.__new__.__defaults__ = (None,)


class MaskSpec(_MaskSpecBase):
    """Per-sample description of the self-attention mask of seq2seq_loader.py:292-301; second_st / second_end / is_s2s: int32 [B].
    second_st = len(tokens_a) + 2, second_end = len(tokens_a) + len(tokens_b) + 3, is_s2s = 1 (seq2seq) / 0 (bidirectional).
    lens_host (optional): second_end as a HOST list of ints -- the number of leading positions of each sample that anything attends.
    The loader knows it for free; with it the engine's padding-free step (VLP_VARLEN=1) needs no device read-back."""
    __slots__ = ()

    @staticmethod
    def from_lengths(len_a, len_b, s2s, device=None):
        """len_a: region placeholders per sample (int or sequence), len_b: caption tokens without the final [SEP], s2s: bool(s)."""
        len_b = torch.as_tensor(len_b, dtype=torch.int32).reshape(-1)
        B = len_b.numel()
        len_a = torch.as_tensor(len_a, dtype=torch.int32).reshape(-1).expand(B)
        s2s = torch.as_tensor(s2s).to(torch.int32).reshape(-1).expand(B)
        end = (len_a + len_b + 3).contiguous()
        spec = MaskSpec((len_a + 2).contiguous(), end, s2s.contiguous(), [int(v) for v in end.tolist()])
        return spec if device is None else spec.to(device)

    def to(self, device, non_blocking=False):
        return MaskSpec(*(t.to(device, non_blocking=non_blocking) for t in self[:3]), self.lens_host)

    def check(self, B, L):
        for t in self[:3]:
            if t.dtype != torch.int32 or t.numel() != B or not t.is_cuda:
                raise RuntimeError("MaskSpec fields must be int32 [%d] tensors on the GPU" % B)

    def dense(self, L):
        """The int64 [B, L, L] mask this spec stands for (host/debug helper; the engine never builds it)."""
        st, en, s2s = (t.to(torch.long).view(-1, 1, 1) for t in self[:3])
        q = torch.arange(L, device=self.second_st.device).view(1, L, 1)
        k = torch.arange(L, device=self.second_st.device).view(1, 1, L)
        tri = (k < st) | ((q >= st) & (q < en) & (k >= st) & (k <= q))
        return torch.where(s2s.bool(), tri, (k < en).expand(-1, L, -1)).to(torch.long)
