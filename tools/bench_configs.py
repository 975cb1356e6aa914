#!/usr/bin/env python3
"""Single-GPU figures for the BASELINE.json configs that are not the bench.py headline (config 2):
  1  FB15k-237 abd, B=64, 16 steps, fp32                              (the reference's CPU-runnable case)
  3h DBpedia50 vocabulary (V=25427), B=1024 abd, 32 steps, bf16       (one GPU's share; head NOT vocab-sharded)
  4  N=128 (64 masks/row), 12 layers, 12 heads (hd=64), B=1024/GPU, 32 steps, bf16
  5  RL rollout: T=0.5, top_p=0.9, 32 steps, 4 trajectories per query, B=2048/GPU, bf16 (custom_grpo2.py:605-634)
  wn WN18RR vocabulary (V=41066: the sampler's rows live in the registered workspace), B=1024 abd, 32 steps, bf16
One JSON line per config: sequences/s and the per-kernel split from HIP events is left to rocprofv3."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,3h,4,5")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--gemm-variant", type=int, default=0, help="dark_dbg_set_gemm_variant (A/B runs)")
    a = ap.parse_args()
    from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, _hip, synth
    _hip.check(_hip.load().dark_dbg_set_gemm_variant(a.gemm_variant))
    dev = torch.device("cuda", 0)
    import warnings
    warnings.filterwarnings("ignore", message="Padding was detected")

    def model_for(dtype, **kw):
        cfg = DreamConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, **kw)
        m = DreamModel_length(cfg)
        m.load_state_dict(synth.make_weights(cfg, seed=0), strict=True)
        return m.to(dev).to(dtype)

    def run(name, model, ids, max_length, **kw):
        gc = DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS,
                                   eos_token_id=synth.END, sep_token_id=synth.SEP, max_length=max_length)
        kw = dict(alg="entropy", alg_temp=0, top_p=0.9, skip_noop_steps=False, **kw)
        torch.manual_seed(2)
        model.diffusion_generate(ids, generation_config=gc, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            out = model.diffusion_generate(ids, generation_config=gc, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        assert out.shape == (ids.shape[0], max_length)
        print(json.dumps({"config": name, "sequences_per_s": round(ids.shape[0] / dt, 1),
                          "ms_per_call": round(dt * 1e3, 1), "batch": ids.shape[0], **{k: v for k, v in kw.items()
                                                                                       if k in ("steps", "temperature")}}),
              flush=True)

    want = a.configs.split(",")
    if "1" in want:
        m = model_for(torch.float32, vocab_size=synth.vocab_size(**synth.FB15K237))
        run("1: FB15k-237 abd B=64 16 steps fp32", m, synth.make_inputs(64, "abd", seed=1).to(dev), 46, steps=16,
            temperature=0.1)
        del m
    if "3h" in want:
        m = model_for(torch.bfloat16, vocab_size=synth.vocab_size(**synth.DBPEDIA50))
        ids = synth.make_inputs(1024, "abd", seed=1, **synth.DBPEDIA50).to(dev)
        run("3 (one GPU, unsharded head): DBpedia50 V=25427 abd B=1024 32 steps bf16", m, ids, 46, steps=32,
            temperature=0.1)
        del m
    if "wn" in want:
        m = model_for(torch.bfloat16, vocab_size=synth.vocab_size(**synth.WN18RR))
        ids = synth.make_inputs(1024, "abd", seed=1, **synth.WN18RR).to(dev)
        run("WN18RR V=41066 abd B=1024 32 steps bf16 (row-workspace sampler)", m, ids, 46, steps=32, temperature=0.1)
        del m
    if "4" in want:
        m = model_for(torch.bfloat16, vocab_size=synth.vocab_size(**synth.FB15K237), num_hidden_layers=12,
                      num_attention_heads=12, num_key_value_heads=12, max_position_embeddings=128)
        g = torch.Generator().manual_seed(1)
        ids = torch.randint(100, 15000, (1024, 127), generator=g)
        ids[:, 0] = synth.BOS
        ids[:, 32:95] = synth.MASK            # 63 + the trailing pad mask = 64 masks per row
        run("4: N=128 64 masks/row 12L 12 heads B=1024 32 steps bf16", m, ids.to(dev), 128, steps=32, temperature=0.1)
        del m
    if "5" in want:
        m = model_for(torch.bfloat16, vocab_size=synth.vocab_size(**synth.FB15K237))
        ids = synth.make_inputs(512, "abd", seed=1).repeat_interleave(4, dim=0).to(dev)
        run("5: RL rollout T=0.5 abd 512 queries x 4 trajectories B=2048 32 steps bf16", m, ids, 46, steps=32,
            temperature=0.5)


if __name__ == "__main__":
    main()
