#!/usr/bin/env python3
"""GRPO forward-only log-prob path on one MI355X: scored completions / second.

Workload: `_get_per_token_logps` as the trainer calls it for the old / reference policy
(akgr/abduction_model/custom_grpo2.py:681-709): --batch completions of 46 tokens, --iters iterations, i.e.
3 * iters * batch perturbed sequences through the default DARK shape in bf16, V = 15080.
Legs: fused (dark_lm_head_logprob: logits never written), unfused (dark_lm_head_rows -> fp32 logits [rows, V] ->
dark_logprob_from_logits, what a line-by-line port would do), head-only timings of both with HIP events.  Prints one
JSON line.  The CPU-oracle leg lives in bench.py (`python bench.py --workload logprob`).
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
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--reps", type=int, default=10)
    return ap.parse_args(argv)


def run(a):
    from dark_amd import DreamConfig, DreamModel_length, _hip, grpo_logps, synth
    dev = torch.device("cuda", 0)
    lib = _hip.load()
    cfg = DreamConfig(vocab_size=synth.vocab_size(**synth.FB15K237), mask_token_id=synth.MASK,
                      pad_token_id=synth.END)
    w = synth.make_weights(cfg, seed=0)
    model = DreamModel_length(cfg)
    model.load_state_dict(w, strict=True)
    model = model.to(dev).to(torch.bfloat16)
    I, B, N, V = a.iters, a.batch, 46, cfg.vocab_size
    g = torch.Generator().manual_seed(0)
    ids = torch.nn.functional.pad(synth.make_inputs(I * B, "abd", seed=1), (0, 1), value=synth.END)
    target = (ids == synth.MASK)
    ids = torch.where(target, torch.randint(100, V, ids.shape, generator=g), ids).view(I, B, N).to(dev)
    target = target.view(I, B, N).to(dev)
    seeds = list(range(1, I + 1))

    def call():
        return grpo_logps.get_per_token_logps(model, ids, target, seeds, synth.MASK)

    out = call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    assert out.shape == (B, I, N) and bool(torch.isfinite(out).all())

    # head-only A/B on the hidden states of the last run
    S = 3 * I * B
    M = S * N
    st = model._hip_state(S, N)
    pk = model.packed()
    src = torch.empty(M, dtype=torch.int32, device=dev)
    s = _hip.stream_ptr(dev)
    _hip.check(lib.dark_shift_rows(_hip.ptr(src), S, N, s))
    targets = ids.view(I, 1, B, N).expand(I, 3, B, N).contiguous().view(-1)
    ws_b = lib.dark_logprob_workspace_bytes(pk.ref, M)
    ws = torch.empty(ws_b, dtype=torch.uint8, device=dev)
    lp = torch.empty(M, dtype=torch.float32, device=dev)
    hws_b = lib.dark_head_workspace_bytes(pk.ref, M)
    hws = torch.empty(hws_b, dtype=torch.uint8, device=dev)
    logits = torch.empty((M, V), dtype=torch.float32, device=dev)
    lp2 = torch.empty(M, dtype=torch.float32, device=dev)

    def fused():
        _hip.check(lib.dark_lm_head_logprob(pk.ref, _hip.ptr(st.hidden), _hip.ptr(src), M, _hip.ptr(targets), M,
                                            _hip.ptr(ws), ws_b, _hip.ptr(lp), s))

    def unfused():
        _hip.check(lib.dark_lm_head_rows(pk.ref, _hip.ptr(st.hidden), _hip.ptr(src), M, _hip.ptr(hws), hws_b,
                                         _hip.ptr(logits), V, s))
        _hip.check(lib.dark_logprob_from_logits(_hip.ptr(logits), V, _hip.ptr(targets), M, V, _hip.ptr(lp2), s))

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    ms_f, ms_u = timed(fused), timed(unfused)
    err = float((lp - lp2).abs().max())
    res = {"workload": f"GRPO old/ref-policy log-probs: {I} iterations x {B} completions x 46 tokens = {S} perturbed "
                       f"sequences, bf16, default DARK shape, V={V}",
           "completions_per_s": round(I * B / dt, 1), "ms_per_call": round(dt * 1e3, 2),
           "head_rows": M, "head_fused_ms": round(ms_f, 3), "head_unfused_ms": round(ms_u, 3),
           "head_fused_tflops": round(2.0 * M * V * cfg.hidden_size / (ms_f * 1e-3) / 1e12, 1),
           "logits_bytes_avoided": M * V * 4 * 2, "fused_vs_unfused_max_abs_diff": err}
    return res, w, cfg, ids.cpu(), target.cpu()


def main():
    res = run(parse())[0]
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
