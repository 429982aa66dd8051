"""CPU checks of the N3 host logic (vlp_amd/data.py): packed store round trip and TextPreprocessor invariants.  The exact
reproduction of the reference's sample stream is pinned in tests/test_oracle_vs_reference.py (needs /root/reference)."""
import random

import numpy as np
import pytest

from oracle import loader_oracle as LO
from vlp_amd.data import PackedRegionStore, TextPreprocessor, pack_from_h5, write_packed


def test_packed_store_round_trip(tmp_path):
    rng = np.random.RandomState(0)
    n, nv = 5, 100
    feats = np.abs(rng.standard_normal((n, nv, 2048))).astype(np.float16)
    cls = rng.rand(n, nv, 1601).astype(np.float16)
    box = rng.rand(n, nv, 6).astype(np.float32)
    ids = ["COCO_%012d" % (7 * i) for i in range(n)]
    write_packed(str(tmp_path), ids, feats, cls, box)
    st = PackedRegionStore(str(tmp_path))
    assert len(st) == n and st.nv == nv
    rows = st.rows([ids[3], ids[0], ids[3]])
    f, c, b = np.empty((3, nv, 2048), np.float16), np.empty((3, nv, 1601), np.float16), np.empty((3, nv, 6), np.float32)
    st.gather(rows, f, c, b)
    assert np.array_equal(f[0], feats[3]) and np.array_equal(f[1], feats[0]) and np.array_equal(c[2], cls[3]) and np.array_equal(b[1], box[0])
    with pytest.raises(ValueError):
        write_packed(str(tmp_path), ids, feats[:, :, :100], cls, box)
    with pytest.raises(KeyError):
        st.rows(["missing"])


def test_pack_from_h5_reports_missing_dependency(tmp_path):
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py present")
    except ImportError:
        with pytest.raises(RuntimeError, match="h5py"):
            pack_from_h5("prefix", "bbox.h5", ["a"], str(tmp_path))


@pytest.mark.parametrize("mode", ["s2s", "bi"])
def test_text_preprocessor_invariants(mode):
    tp = TextPreprocessor(3, 0.15, 1000, cls_id=101, sep_id=102, mask_id=103, unk_id=100, max_len=123, max_len_b=20, mode=mode)
    random.seed(3)
    for n in (1, 7, 20, 33):
        toks = list(range(200, 200 + n))
        t = tp(toks)
        nb = min(n, 20)
        assert len(t["input_ids"]) == 123 and t["input_ids"][0] == 101 and t["input_ids"][101] == 102 and t["len_b"] == nb
        assert t["input_ids"][1:101] == [100] * 100 and t["input_ids"][102 + nb] in (102, 103) or t["input_ids"][102 + nb] < 1000
        assert t["segment_ids"] == LO.segment_ids(100, nb, 123, mode).tolist()
        npred = sum(t["masked_weights"])
        assert npred == min(3, max(1, int(round(nb * 0.15)))) and all(102 <= p <= 102 + nb for p in t["masked_pos"][:npred])
        # the labels are the ORIGINAL tokens at the masked positions
        orig = [101] + [100] * 100 + [102] + toks[:nb] + [102]
        assert [orig[p] for p in t["masked_pos"][:npred]] == t["masked_ids"][:npred]


@pytest.mark.parametrize("n,world", [(103, 8), (64, 2), (5, 8), (1000, 3)])
def test_distributed_sampler_indices_equal_torch_distributed_sampler(n, world):
    """The reference shards with torch's DistributedSampler and calls set_epoch every epoch (run_img2txt_dist.py:295, 455): ONE global
    permutation per epoch, rank r takes every world-th element.  vlp_amd.data.distributed_sampler_indices must yield the same indices
    (no process group needed: num_replicas / rank are given), cover the dataset across the ranks, and change with the epoch."""
    from torch.utils.data.distributed import DistributedSampler

    from vlp_amd.data import distributed_sampler_indices
    data = list(range(n))
    for epoch in (0, 1, 7):
        seen = []
        for rank in range(world):
            ref = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True, seed=0)
            ref.set_epoch(epoch)
            mine = distributed_sampler_indices(n, world, rank, epoch)
            assert mine == list(iter(ref)) and len(mine) == -(-n // world)
            seen += mine
        assert set(seen) == set(data)
    assert distributed_sampler_indices(n, world, 0, 0) != distributed_sampler_indices(n, world, 0, 1) or n <= world
