"""mhim_mil_amd/masking.py (the reference's free functions select_mask_fn / mask_fn, masking.py:9-110) against the reference's own
outputs (tests/golden g5_*) and the oracle: both multi-head fusions, select_inv, the union with an earlier mask."""
import numpy as np
import pytest
import torch

from mhim_mil_amd import synth
from oracle import mhim_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


def test_mean_fusion_matches_the_reference():
    from mhim_mil_amd.masking import select_mask_fn
    meta, a = G.load("g5_select_mean_n600")
    lk, ids = select_mask_fn(meta["n"], _t(a["attn"])[None], True, meta["mask_ratio_h"], len_keep_other=meta["n"],
                             random_ratio=meta["mask_ratio_hr"], msa_fusion="mean", perm=a["perm"])
    assert lk == int(a["len_keep"]) and ids.shape == (1, meta["n"])
    ids = ids[0].cpu().numpy()
    assert np.array_equal(ids[:lk], a["kept"]) and np.array_equal(ids[lk:], a["masked"])


def test_select_inv_matches_the_reference():
    from mhim_mil_amd.masking import select_mask_fn
    meta, a = G.load("g5_select_inv_n512")
    lk, ids = select_mask_fn(meta["n"], _t(a["score"])[None], True, meta["mask_ratio_h"], select_inv=True)
    ids = ids[0].cpu().numpy()
    assert lk == int(a["len_keep"])
    assert np.array_equal(np.sort(ids[:lk]), np.sort(a["first"])) and np.array_equal(ids[lk:], a["rest"])
    assert np.array_equal(ids[:lk], a["first"])              # tie-free scores: torch.topk's order is value-descending, as ours


def test_two_d_and_vote_match_the_reference_fixtures():
    from mhim_mil_amd.masking import select_mask_fn
    meta, a = G.load("g5_select_low_n512")
    lk, ids = select_mask_fn(meta["n"], _t(a["score"])[None], False, meta["mask_ratio_l"])
    ids = ids[0].cpu().numpy()
    assert lk == int(a["len_keep"]) and np.array_equal(ids[:lk], a["kept"]) and np.array_equal(np.sort(ids[lk:]), np.sort(a["masked"]))
    meta, a = G.load("g5_select_vote_n600")
    lk, ids = select_mask_fn(meta["n"], _t(a["attn"])[None], True, meta["mask_ratio_h"], len_keep_other=meta["n"],
                             random_ratio=meta["mask_ratio_hr"], msa_fusion="vote", perm=a["perm"])
    ids = ids[0].cpu().numpy()
    assert lk == int(a["len_keep"]) and len(set(ids.tolist())) == meta["n"] and np.all(np.diff(ids[:lk]) > 0)


@pytest.mark.parametrize("fusion", ["mean", "vote"])
@pytest.mark.parametrize("inv", [False, True])
def test_options_against_the_oracle_with_an_earlier_mask(fusion, inv):
    """The union with an earlier mask (masking.py:36-39,74-75) under both fusions and select_inv: tie-free per-head scores."""
    from mhim_mil_amd.masking import select_mask_fn
    n, h = 900, 8
    a = ((np.stack([synth.permutation(300 + i, n) for i in range(h)]) + 0.25 * synth.uniform(91, (h, n))) / n).astype(np.float32)
    other = np.sort(synth.permutation(17, n)[:60]).astype(np.int64)
    ratio, rr = 0.12, 0.5
    k = int(np.ceil(n * ratio / rr))
    if fusion == "mean":
        kk = k // h
        kc = len(np.unique(np.concatenate([O.topk_indices(a[i], kk, True) for i in range(h)])))
    else:
        kc = k
    perm = synth.permutation(23, kc)
    lk_o, ids_o, _ = O.select_mask(n, a, True, ratio, other_masked=other, random_ratio=rr, perm=perm, select_inv=inv, msa_fusion=fusion)
    rest = np.setdiff1d(np.arange(n), other)
    prev = _t(np.concatenate([rest, other]))[None]            # the earlier call's mask_ids: kept ++ masked (mhim.py:136-139 passes both)
    lk, ids = select_mask_fn(n, _t(a)[None], True, ratio, mask_ids_other=prev, len_keep_other=len(rest), cls_attn_topk_idx_other=_t(other),
                             random_ratio=rr, select_inv=inv, msa_fusion=fusion, perm=perm)
    # (the earlier mask derived from mask_ids_other alone shrinks ps for the ratio, masking.py:37-40: the oracle's shrink_ps)
    lk2_o, ids2_o, _ = O.select_mask(n, a[0], True, ratio, other_masked=other, shrink_ps=True)
    lk2, ids2 = select_mask_fn(n, _t(a[0])[None], True, ratio, mask_ids_other=prev, len_keep_other=len(rest))
    assert lk2 == lk2_o and np.array_equal(ids2[0].cpu().numpy()[:lk2], ids2_o[:lk2_o])
    ids = ids[0].cpu().numpy()
    assert lk == lk_o
    if fusion == "mean":                                      # candidate order is fixed (ascending union): exact
        assert np.array_equal(ids, ids_o)
    else:                                                     # votes tie structurally: the sets obey the contract, not one order
        assert len(set(ids.tolist())) == n and set(other.tolist()) <= set((ids[:lk] if inv else ids[lk:]).tolist())


def test_mask_fn_gathers_the_kept_rows():
    from mhim_mil_amd.masking import mask_fn
    x = torch.randn(1, 700, 96, device=DEV)
    ids = _t(synth.permutation(5, 700).astype(np.int64))[None]
    out = mask_fn(x, ids, 333)
    assert out.shape == (1, 333, 96) and torch.equal(out[0], x[0][ids[0, :333]])
