"""Rank process of tests/test_hip_dispatch.py (not collected by pytest): the REAL model through
dark_amd.dispatch.generate_sharded, world_size ranks sharing cuda:0, collectives over gloo.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment;  argv[1] = output .pt path prefix
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, dispatch, synth

KW = dict(vocab_size=15080, hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=8,
          num_key_value_heads=8, max_position_embeddings=64, mask_token_id=synth.MASK, pad_token_id=synth.END)


def build(device="cuda:0"):
    cfg = DreamConfig(**KW)
    m = DreamModel_length(cfg)
    m.load_state_dict(synth.make_weights(cfg, seed=0), strict=True)
    return m.to(device)


def ragged_inputs():
    """12 abduction rows whose mask counts differ (the batch-average k couples the shards, generation_utils.py:520)."""
    ids = synth.make_inputs(12, "abd", seed=17)
    ids[0, 1:8] = torch.arange(201, 208)     # 3 + 1 masks
    ids[3, 1:10] = torch.arange(211, 220)    # 1 + 1 masks
    ids[7, 20:30] = synth.MASK               # 21 masks
    ids[11, 1:11] = torch.arange(221, 231)   # only the padded mask
    return ids


def gen_config():
    return DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS,
                                 eos_token_id=synth.END, sep_token_id=synth.SEP, max_length=46)


def shard_mode(rank, world, prefix):
    """Data-parallel rows + the vocabulary-sharded head (BASELINE config 3) against the plain data-parallel run."""
    from dark_amd.vocab_shard import ShardedHead
    m = build()
    ids = synth.make_inputs(12, "ded", seed=23).to("cuda:0")
    kw = dict(steps=10, temperature=0.5, alg="entropy", alg_temp=0, top_p=0.9)
    res = {}
    torch.manual_seed(300 + rank)
    res["plain"] = dispatch.generate_sharded(m, ids, gen_config(), **kw).cpu()
    torch.manual_seed(300 + rank)
    head = ShardedHead(m, None)
    lo, hi = dispatch.shard_bounds(12, world, rank)
    local = m.diffusion_generate(ids[lo:hi], generation_config=gen_config(), _sharded_head=head, **kw)
    res["sharded"] = dispatch.gather_rows(local, 12).cpu()
    torch.manual_seed(300 + rank)
    again = dispatch.generate_sharded(m, ids, gen_config(), shard_head=True, **kw).cpu()
    assert torch.equal(again, res["sharded"])
    res["fallback"] = (head.fallback_rows, head.rows)
    torch.save(res, f"{prefix}.rank{rank}.pt")


def nccl_single_rank(prefix):
    """ONE rank on cuda:0 over `nccl` (= RCCL) with dispatch.FORCE_COLLECTIVES: every collective of the multi-GPU
    path (all_gather of sequences / records / hidden rows / flags, all_reduce SUM of the mask count and MAX of the row
    maxima, all_to_all of the shard records, barrier) runs through RCCL on device tensors of the dtypes the real run
    uses -- the only way to take those branches on a 1-GPU box -- and must leave the results unchanged."""
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    m = build()
    rag = ragged_inputs().to("cuda:0")
    kw = dict(steps=8, temperature=0.0, alg="entropy", alg_temp=0, top_p=0.9)
    res = {"plain_ragged": m.diffusion_generate(rag, generation_config=gen_config(), **kw).cpu()}
    ids = synth.make_inputs(12, "ded", seed=23).to("cuda:0")
    kw5 = dict(steps=10, temperature=0.5, alg="entropy", alg_temp=0, top_p=0.9)
    torch.manual_seed(300)
    res["plain_t05"] = m.diffusion_generate(ids, generation_config=gen_config(), **kw5).cpu()
    dispatch.FORCE_COLLECTIVES = True
    res["nccl_ragged"] = dispatch.generate_sharded(m, rag, gen_config(), exact_ragged_k=True, **kw).cpu()
    torch.manual_seed(300)
    res["nccl_shard_head"] = dispatch.generate_sharded(m, ids, gen_config(), shard_head=True, **kw5).cpu()
    bf = build().to(torch.bfloat16)          # bf16 hidden rows through the all_gather
    torch.manual_seed(301)
    res["plain_bf16"] = bf.diffusion_generate(ids, generation_config=gen_config(), **kw5).cpu()
    torch.manual_seed(301)
    res["nccl_bf16_shard_head"] = dispatch.generate_sharded(bf, ids, gen_config(), shard_head=True, **kw5).cpu()
    scores = torch.arange(6, dtype=torch.float32).view(3, 2)
    pid = torch.tensor([1, 2, 3], dtype=torch.int32)
    sc, pi = dispatch.gather_metric_records(scores.to("cuda:0"), pid.to("cuda:0"), 3)
    res["scores"], res["pattern_id"] = sc.cpu(), pi.cpu()
    t = torch.tensor([1.5, 2.5], device="cuda:0", dtype=torch.float64)   # bench.py's max-over-ranks of the timings
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    res["f64_max"] = t.cpu()
    dist.barrier()
    torch.cuda.synchronize()
    torch.save(res, f"{prefix}.rank0.pt")
    dist.destroy_process_group()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if len(sys.argv) > 2 and sys.argv[2] == "nccl1":
        return nccl_single_rank(sys.argv[1])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    if len(sys.argv) > 2 and sys.argv[2] == "shard":
        shard_mode(rank, world, sys.argv[1])
        dist.barrier()
        dist.destroy_process_group()
        return
    m = build()
    rag = ragged_inputs().to("cuda:0")
    res = {}
    kw = dict(steps=8, temperature=0.0, alg="entropy", alg_temp=0, top_p=0.9)
    res["ragged_exact"] = dispatch.generate_sharded(m, rag, gen_config(), exact_ragged_k=True, **kw).cpu()
    res["ragged_local_k"] = dispatch.generate_sharded(m, rag, gen_config(), exact_ragged_k=False, **kw).cpu()
    uni = synth.make_inputs(10, "ded", seed=5).to("cuda:0")
    torch.manual_seed(100 + rank)
    res["uniform_t05"] = dispatch.generate_sharded(m, uni, gen_config(), steps=8, temperature=0.5, alg="entropy",
                                                   alg_temp=0, top_p=0.9).cpu()
    scores = torch.arange(6, dtype=torch.float32).view(3, 2) + 10 * rank
    pid = torch.tensor([1, 2, 3], dtype=torch.int32) + 10 * rank
    sc, pi = dispatch.gather_metric_records(scores.to("cuda:0"), pid.to("cuda:0"), 3 * world)
    res["scores"], res["pattern_id"] = sc.cpu(), pi.cpu()
    torch.save(res, f"{sys.argv[1]}.rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
