#!/usr/bin/env python3
"""Throughput of the variable-length EXPAND / DELETE sampler on one MI355X.

Workload = what `main.test` feeds it (akgr/abduction_model/main.py:236-274, :795): abduction prompts
`[BOS, MASK x5, SEP, observation, EOS]`, max_length 50, T = 0.1, top_p = 0.9, alg 'entropy', one token per step;
default DARK shape, bf16, random-init weights with the EXPAND / DELETE head rows boosted so that the edit branches
run (a plain random-init model never emits them).  Three legs:
  batched : diffusion_generate_batch over --batch ragged prompts (this repo's reformulation)
  single  : the reference's call pattern, one [1, L] prompt per diffusion_generate call, on the same GPU path
Prints one JSON line.  The CPU-oracle leg lives in bench.py (`python bench.py --workload expand`), the only
benchmark allowed to touch oracle/.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--masks", type=int, default=5)
    ap.add_argument("--single", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--gains", type=float, nargs=2, default=(8.0, 7.0))
    ap.add_argument("--no-compact", action="store_true")
    return ap.parse_args(argv)


def run(a):
    """GPU legs; returns (result dict, weights, config, prompts on the CPU) for a caller that adds a CPU baseline."""
    from dark_amd import DreamConfig, DreamModel, synth
    from dark_amd.generation_expand_utils import DreamGenerationConfig
    dev = torch.device("cuda", 0)
    cfg = DreamConfig(vocab_size=synth.vocab_size(**synth.FB15K237), mask_token_id=synth.MASK,
                      pad_token_id=synth.END)
    w = synth.boost_edit_tokens(synth.make_weights(cfg, seed=0), *a.gains)
    model = DreamModel(cfg)
    model.load_state_dict(w, strict=True)
    model = model.to(dev).to(torch.bfloat16)
    gc = DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS,
                               eos_token_id=synth.END, expand_token_id=synth.EXPAND, delete_token_id=synth.END,
                               max_length=50)
    kw = dict(temperature=0.1, alg="entropy", alg_temp=0, top_p=0.9, number_transfer_tokens=1)
    prompts = [p.to(dev) for p in synth.make_var_inputs(a.batch, "abd", a.masks, seed=5)]
    torch.manual_seed(0)

    trace = {}
    model.diffusion_generate_batch(prompts, generation_config=gc, compact_finished=not a.no_compact, _trace=trace,
                                   **kw)  # warm-up + step statistics
    steps = len(trace["M"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = model.diffusion_generate_batch(prompts, generation_config=gc, compact_finished=not a.no_compact, **kw)
    torch.cuda.synchronize()
    dt_b = (time.perf_counter() - t0) / a.reps
    assert out.shape == (a.batch, 50)

    model.diffusion_generate(inputs=prompts[0], generation_config=gc, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in prompts[:a.single]:
        model.diffusion_generate(inputs=p, generation_config=gc, **kw)
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0

    res = {"workload": f"variable-length sampler, {a.batch} abd prompts x {a.masks} masks, max_length 50, T=0.1 "
                       "top_p=0.9 entropy k=1, bf16, default DARK shape",
           "batched_seq_per_s": round(a.batch / dt_b, 1), "batched_ms": round(dt_b * 1e3, 1),
           "loop_iterations": steps, "masked_rows_total": int(sum(trace["M"])),
           "single_seq_per_s": round(a.single / dt_s, 2), "single_ms_per_seq": round(dt_s / a.single * 1e3, 2),
           "compaction": not a.no_compact}
    return res, w, cfg, [p.cpu() for p in prompts]


def main():
    res, _, _, _ = run(parse())
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
