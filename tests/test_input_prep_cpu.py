"""CPU checks of the N2 host logic: the loader oracle against the fixtures the unmodified reference pipeline produced
(tests/golden/loader_*.npz) and vlp_amd.input_prep.MaskSpec against the oracle's mask rules."""
import os

import numpy as np
import pytest
import torch

from oracle import loader_oracle as LO
from oracle.make_golden import LOADER_CASES
from vlp_amd.input_prep import MaskSpec, RawRegions

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", list(LOADER_CASES.keys()))
def test_loader_oracle_vs_reference_fixture(name):
    mode = LOADER_CASES[name][0]
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    nb, L = int(g["len_b"]), g["input_mask"].shape[0]
    assert np.array_equal(g["input_mask"], LO.attention_mask(100, nb, L, mode))
    assert np.array_equal(g["segment_ids"], LO.segment_ids(100, nb, L, mode))
    assert np.abs(LO.vis_pe_prepare(g["bbox"], g["cls"].astype(np.float32)) - g["vis_pe"]).max() < 2e-4


def test_mask_spec_dense_matches_oracle():
    len_b = [1, 9, 20, 14]
    modes = ["s2s", "s2s", "bi", "s2s"]
    spec = MaskSpec.from_lengths(100, len_b, [m == "s2s" for m in modes])
    L = 123
    dense = spec.dense(L).numpy()
    for b, (nb, m) in enumerate(zip(len_b, modes)):
        assert np.array_equal(dense[b], LO.attention_mask(100, nb, L, m)), (b, nb, m)
    assert spec.second_st.dtype == torch.int32 and spec.second_end.tolist() == [104, 112, 123, 117]


def test_raw_regions_shape_and_validation():
    r = RawRegions(torch.zeros(2, 100, 6), torch.zeros(2, 100, 1601, dtype=torch.float16))
    assert r.shape == (2, 100, 1607)
    with pytest.raises(RuntimeError):
        r.check(2, 100)                         # CPU tensors: there is no CPU path
    with pytest.raises(RuntimeError):
        RawRegions(torch.zeros(2, 100, 5), torch.zeros(2, 100, 1601)).check(2, 100)
