"""BASELINE config 3: the vocabulary-sharded head + sampler (dark_amd/vocab_shard.py, sample.hip) must give the SAME
bits as the unsharded path.  Proven on one GPU by running the S column shards in sequence (kernel level on planted
logits, then through the whole loop at V = 25427), and with two rank processes sharing cuda:0 over gloo."""
import os
import sys

import numpy as np
import pytest
import torch

from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, _hip, synth
from dark_amd.generation_utils import philox_fill_plan
from dark_amd.vocab_shard import ShardedHead, shard_bounds

pytestmark = pytest.mark.gpu
DEV = "cuda"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _logits(M, V, seed):
    g = torch.Generator().manual_seed(seed)
    lg = torch.randn(M, V, generator=g) * 0.6
    lg[: M // 2] *= 9.0                       # peaked rows: the head path
    lg[3, :40] = lg[3, 0]                     # a run of equal logits at the top of a peaked row
    lg[5] = 0.0                               # flat row (every shard's record overflows)
    n7 = lg[7, ::97].numel()                  # > 64 near-equal head candidates spread over the shards
    lg[7, ::97] = 30.0 + 1e-4 * torch.arange(n7)
    lg[9, 100] = 50.0                         # a single dominant token
    return lg


def _sharded(lib, lg_dev, S, T, top_p, top_k, mode, noise, philox):
    M, V = lg_dev.shape
    st = torch.cuda.current_stream().cuda_stream
    mx = torch.full((M,), float("-inf"), device=DEV)
    shards = []
    for s in range(S):
        v0, Vs = shard_bounds(V, S, s)
        if Vs == 0:
            shards.append((None, v0, 0))
            continue
        sl = lg_dev[:, v0:v0 + Vs].contiguous()  # what rank s would hold
        pm = torch.empty(M, device=DEV)
        _hip.check(lib.dark_shard_row_max(_hip.ptr(sl), sl.stride(0), M, Vs, T, _hip.ptr(pm), st))
        mx = torch.maximum(mx, pm)
        shards.append((sl, v0, Vs))
    parts = torch.zeros((S, M, _hip.SHARD_PART_BYTES), dtype=torch.uint8, device=DEV)
    for s, (sl, v0, Vs) in enumerate(shards):
        if Vs:
            _hip.check(lib.dark_shard_partials(_hip.ptr(sl), sl.stride(0), M, Vs, v0, T, _hip.ptr(mx), _hip.ptr(parts[s]), st))
    conf = torch.full((M,), float("nan"), device=DEV)
    x0 = torch.full((M,), -1, dtype=torch.long, device=DEV)
    flags = torch.full((M,), 7, dtype=torch.uint8, device=DEV)
    seed, off, nth = philox if philox else (0, 0, 0)
    _hip.check(lib.dark_sample_from_shards(_hip.ptr(parts), M, S, M, V, T, top_p, top_k, mode, _hip.ptr(mx),
                                           _hip.ptr(noise), V if noise is not None else 0, seed, off, nth,
                                           _hip.ptr(conf), _hip.ptr(x0), _hip.ptr(flags), st))
    torch.cuda.synchronize()
    assert bool(((flags == 0) | (flags == 1)).all())
    rows = flags.nonzero().flatten().to(torch.int32)
    if rows.numel():
        sub = lg_dev[rows.long()].contiguous()
        _hip.check(lib.dark_sample_rows_indexed(_hip.ptr(sub), V, rows.numel(), V, T, top_p, top_k, mode, _hip.ptr(rows),
                                                _hip.ptr(noise), V if noise is not None else 0, seed, off, nth,
                                                _hip.ptr(conf), _hip.ptr(x0), st))
        torch.cuda.synchronize()
    return conf.cpu(), x0.cpu(), flags.cpu()


@pytest.mark.parametrize("V", [15080, 25427])
@pytest.mark.parametrize("T,top_p,top_k,mode", [(0.1, 0.9, 0, 2), (0.5, 0.9, 0, 2), (0.0, 0.9, 0, 2), (0.3, 0.95, 0, 0),
                                                 (0.7, 0.8, 0, 1), (0.5, -1.0, 0, 2), (0.5, 0.9, 20, 2)])
def test_sharded_sampler_is_bit_identical(V, T, top_p, top_k, mode):
    lib = _hip.load()
    M = 96
    lg = _logits(M, V, V + int(T * 10)).to(DEV)
    g = torch.Generator().manual_seed(1)
    noise = torch.empty(M, V).exponential_(1, generator=g).to(DEV) if T > 0 else None
    st = torch.cuda.current_stream().cuda_stream
    for use_philox in ((False, True) if T > 0 else (False,)):
        philox = None
        if use_philox:
            torch.manual_seed(5)
            philox = philox_fill_plan(torch.device(DEV), M * V)
        ref_c = torch.empty(M, device=DEV)
        ref_x = torch.empty(M, dtype=torch.long, device=DEV)
        if philox:
            _hip.check(lib.dark_sample_rows_philox(_hip.ptr(lg), V, M, V, T, top_p, top_k, mode, *philox, _hip.ptr(ref_c),
                                                   _hip.ptr(ref_x), st))
        else:
            _hip.check(lib.dark_sample_rows(_hip.ptr(lg), V, M, V, T, top_p, top_k, mode, _hip.ptr(noise),
                                            V if noise is not None else 0, _hip.ptr(ref_c), _hip.ptr(ref_x), st))
        torch.cuda.synchronize()
        for S in (1, 3, 8):
            conf, x0, flags = _sharded(lib, lg, S, T, top_p, top_k, mode, None if philox else noise, philox)
            assert torch.equal(x0, ref_x.cpu()), (S, use_philox, (x0 != ref_x.cpu()).nonzero().flatten().tolist())
            assert torch.equal(conf.view(torch.int32), ref_c.cpu().view(torch.int32)), (S, use_philox)
            nucleus = 0 <= top_p < 1 and top_k == 0
            if nucleus:
                assert int(flags[5]) == 1 and int(flags[7]) == 1      # flat / too many candidates: gathered rows
                if T > 0:
                    assert int(flags[: M // 2].sum()) <= M // 4, "most peaked rows must be finished from the shard records"
            else:
                assert bool((flags == 1).all())                       # no nucleus / top-k: every row is gathered


KW = dict(vocab_size=25427, hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=8,
          num_key_value_heads=8, max_position_embeddings=64, mask_token_id=synth.MASK, pad_token_id=synth.END)
DBPEDIA = dict(nentity=24624, nrelation=702)


def _gc():
    return DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS,
                                 eos_token_id=synth.END, sep_token_id=synth.SEP, max_length=46)


@pytest.mark.parametrize("dtype,T", [(torch.float32, 0.1), (torch.bfloat16, 0.1), (torch.bfloat16, 0.5)])
def test_loop_with_emulated_shards_equals_unsharded(dtype, T):
    """The whole denoise loop at config 3's vocabulary with the head split into 8 column shards (run in sequence on
    this GPU): histories and the generator's final offset equal the unsharded loop's, lazy Philox noise."""
    cfg = DreamConfig(**KW)
    m = DreamModel_length(cfg)
    m.load_state_dict(synth.make_weights(cfg, seed=0), strict=True)
    m = m.to(DEV).to(dtype)
    ids = synth.make_inputs(48, "ded", seed=8, **DBPEDIA).to(DEV)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    res = []
    for head in (None, ShardedHead(m, n_shards=8), ShardedHead(m, n_shards=3)):
        torch.manual_seed(21)
        out = m.diffusion_generate(ids, generation_config=_gc(), return_dict_in_generate=True, output_history=True,
                                   steps=12, temperature=T, alg="entropy", alg_temp=0, top_p=0.9, _sharded_head=head)
        res.append((torch.stack(out.history).cpu(), gen.get_offset()))
        if head is not None:
            print(f"S={head.S} {dtype} T={T}: {head.fallback_rows} of {head.rows} rows needed their logits gathered")
    for h, off in res[1:]:
        assert torch.equal(h, res[0][0]) and off == res[0][1]


def test_sharded_head_world2(tmp_path):
    """Two rank processes on cuda:0 over gloo (tests/_dispatch_worker.py, mode 'shard'): data-parallel rows AND the
    vocabulary-sharded head; every rank's gathered result equals the plain data-parallel result."""
    from test_hip_dispatch import _run_ranks
    res = _run_ranks(2, str(tmp_path / "out"), mode="shard")
    assert torch.equal(res[0]["sharded"], res[1]["sharded"])
    assert torch.equal(res[0]["sharded"], res[0]["plain"]), "sharded head must not change a single token"
    assert res[0]["fallback"][1] > 0
