"""world_size-2 gloo tests (CPU) of the multi-GPU dispatcher: row sharding, the padded all_gather of results,
fixed-width metric records and the all-reduced mask count behind `exact_ragged_k`."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dark_amd import dispatch, synth
from dark_amd.generation_utils import transfer_count


def test_shard_bounds_cover_rows():
    for total in (0, 1, 7, 64, 1024, 1025):
        for world in (1, 2, 3, 8):
            blocks = [dispatch.shard_bounds(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        dispatch.shard_bounds(4, 2, 2)


class _FakeModel:
    """Stands in for DreamModel_length on CPU: 'denoises' by writing row-dependent ids, so the test can see which
    rows each rank processed and that kwargs arrive unchanged."""

    def diffusion_generate(self, inputs, generation_config=None, **kw):
        assert kw["return_dict_in_generate"] is False and kw["steps"] == 3
        x = torch.nn.functional.pad(inputs, (0, 1), value=synth.MASK)
        x[x == synth.MASK] = 1000 + inputs[:, 12:13].expand_as(x)[x == synth.MASK] % 7
        if "_count_reduce" in kw:
            m, b = kw["_count_reduce"](torch.tensor([int((inputs == synth.MASK).sum())]), inputs.shape[0])
            x[:, -1] = m * 1000 + b
        return x

    def diffusion_generate_batch(self, prompts, generation_config=None, **kw):
        assert kw["return_dict_in_generate"] is False
        n = generation_config.max_length
        out = torch.full((len(prompts), n), synth.END)
        for b, p in enumerate(prompts):
            out[b, :p.shape[1]] = torch.where(p[0] == synth.MASK, 2000 + p.shape[1], p[0])
        return out


def _worker(rank, world, port, total):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ids = synth.make_inputs(total, "abd", seed=3)
        ids[0, 1:6] = 200  # ragged: row 0 has fewer masks
        model = _FakeModel()
        out = dispatch.generate_sharded(model, ids, steps=3)
        exp = model.diffusion_generate(ids, return_dict_in_generate=False, steps=3)
        assert torch.equal(out, exp), "sharded + gathered result differs from the unsharded call"
        local = dispatch.generate_sharded(model, ids, steps=3, gather=False)
        lo, hi = dispatch.shard_bounds(total, world, rank)
        assert torch.equal(local, exp[lo:hi])
        # exact_ragged_k: every rank sees the global (mask count, batch) pair
        out = dispatch.generate_sharded(model, ids, steps=3, exact_ragged_k=True)
        m_glob = int((ids == synth.MASK).sum())
        assert bool((out[:, -1] == m_glob * 1000 + total).all())
        ts = torch.linspace(1, 1e-3, 5)
        assert transfer_count(m_glob, total, ts[0], ts[1], False) == int(torch.tensor(m_glob) / total * (1 - ts[1] / ts[0]))
        # variable-length path: ragged prompt list, contiguous blocks, gathered [total, max_length]
        import types
        prompts = synth.make_var_inputs(total, "abd", 5, seed=9)
        gc = types.SimpleNamespace(max_length=50)
        got = dispatch.generate_batch_sharded(model, prompts, gc)
        assert torch.equal(got, model.diffusion_generate_batch(prompts, gc, return_dict_in_generate=False))
        # metric records
        g = torch.Generator().manual_seed(rank)
        scores = torch.rand(hi - lo, 6, generator=g)
        pid = torch.arange(lo, hi, dtype=torch.int32) % 13
        s_all, p_all = dispatch.gather_metric_records(scores, pid, total)
        assert s_all.shape == (total, 6) and torch.equal(p_all, torch.arange(total, dtype=torch.int32) % 13)
        assert torch.equal(s_all[lo:hi], scores)
        other = 1 - rank
        olo, ohi = dispatch.shard_bounds(total, world, other)
        assert torch.equal(s_all[olo:ohi], torch.rand(ohi - olo, 6, generator=torch.Generator().manual_seed(other)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 9])
def test_generate_sharded_gloo_world2(total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, total), nprocs=2, join=True)
