"""Layout build / SEP-split decode: the numpy oracle against fixtures produced by the reference's own
definitions and tokenizer (CPU), and the device kernels against both (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import layout_oracle as L

HERE = os.path.dirname(os.path.abspath(__file__))
STRIP = (0, 2, 3, 4, 5, 6)   # START END UNK SEP EXPAND MASK: the tokens the reference registers as special


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "reference_goldens_layout.json")) as f:
        meta = json.load(f)
    return np.load(os.path.join(HERE, "golden", "reference_goldens_layout.npz")), meta


def _strings(meta, lists):
    return [" ".join(meta["id_to_token"][t] for t in row) for row in lists]


def test_oracle_build_layout(gold):
    g, meta = gold
    i = meta["ids"]
    for mt in ("unify", "sft_abd", "sft_ded"):
        ids, am = L.build_layout(g["query_ids"], g["observation_ids"], mt, i["bos"], i["sep"], i["eos"], i["mask"])
        assert np.array_equal(ids, g[f"length_{mt}_ids"]), mt
        assert np.array_equal(am, g[f"length_{mt}_mask"].astype(np.float32)), mt
    for mt in ("sft_abd", "sft_ded"):
        for b in range(4):
            got = L.build_single(g[f"single_{mt}_{b}_side"], 5, mt, i["bos"], i["sep"], i["eos"], i["mask"])
            assert np.array_equal(got, g[f"single_{mt}_{b}_ids"]), (mt, b)


def test_oracle_split_strip(gold):
    g, meta = gold
    sep_pos, left, right = L.split_strip(g["post_ids"], meta["ids"]["sep"], STRIP)
    assert sep_pos == meta["post_sep_pos"]
    assert _strings(meta, left) == meta["post_left"]
    assert _strings(meta, right) == meta["post_right"]
    with pytest.raises(IndexError):
        L.split_strip(np.full((2, 8), 101), meta["ids"]["sep"], STRIP)


@pytest.mark.gpu
def test_device_build_layout(gold):
    from dark_amd import layout
    g, meta = gold
    i = meta["ids"]
    q, o = torch.from_numpy(g["query_ids"]).cuda(), torch.from_numpy(g["observation_ids"]).cuda()
    for mt in ("unify", "sft_abd", "sft_ded"):
        ids, need = layout.build_inputs(q, o, mt, i["bos"], i["sep"], i["eos"], i["mask"])
        assert np.array_equal(ids.cpu().numpy(), g[f"length_{mt}_ids"]), mt
        assert np.array_equal(need.cpu().numpy(), g[f"length_{mt}_mask"].astype(np.float32)), mt
    # a larger ragged-free batch against the oracle
    gen = torch.Generator().manual_seed(0)
    q = torch.randint(0, 500, (1000, 10), generator=gen)
    o = torch.randint(0, 500, (1000, 32), generator=gen)
    for mt in ("unify", "sft_abd", "sft_ded"):
        ids, need = layout.build_inputs(q.cuda(), o.cuda(), mt, 0, 4, 2, 6)
        wi, wn = L.build_layout(q.numpy(), o.numpy(), mt, 0, 4, 2, 6)
        assert np.array_equal(ids.cpu().numpy(), wi) and np.array_equal(need.cpu().numpy(), wn)
    with pytest.raises(ValueError):
        layout.build_inputs(q.cuda(), o.cuda(), "nope", 0, 4, 2, 6)


@pytest.mark.gpu
def test_device_split_strip(gold):
    from dark_amd import layout
    g, meta = gold
    sep_pos, left, ll, right, rl = layout.split_strip(torch.from_numpy(g["post_ids"]).cuda(), meta["ids"]["sep"], STRIP)
    assert sep_pos == meta["post_sep_pos"]
    assert _strings(meta, layout.to_lists(left, ll)) == meta["post_left"]
    assert _strings(meta, layout.to_lists(right, rl)) == meta["post_right"]
    assert int((left.cpu()[0, int(ll[0]):] != -1).sum()) == 0
    # random rows, N > 64, against the oracle
    gen = torch.Generator().manual_seed(1)
    x = torch.randint(0, 40, (257, 150), generator=gen)
    x[0, 70] = 4
    x[0, :70][x[0, :70] == 4] = 9
    sp, left, ll, right, rl = layout.split_strip(x.cuda(), 4, STRIP)
    wsp, wl, wr = L.split_strip(x.numpy(), 4, STRIP)
    assert sp == wsp == 70
    assert layout.to_lists(left, ll) == wl and layout.to_lists(right, rl) == wr
    with pytest.raises(IndexError):
        layout.split_strip(torch.full((2, 8), 101).cuda(), 4, STRIP)
