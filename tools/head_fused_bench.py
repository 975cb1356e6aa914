#!/usr/bin/env python3
"""LM head + sampler at the bench shape (M masked rows, V = 15080, bf16): dark_head_sample_rows against the two-call
form, for logits as flat as random-init weights give (peak = 1) and as peaked as a trained model's (final-norm weight
scaled).  Prints the time of both forms and the share of rows that needed their logits.
Usage (GPU box): python tools/head_fused_bench.py [--rows 16896]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dark_amd import DreamConfig, DreamModel_length, _hip, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16896)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--peaks", default="1,3,6,12")
    a = ap.parse_args()
    dev = torch.device("cuda")
    B = (a.rows + 45) // 46
    for peak in [float(v) for v in a.peaks.split(',')]:
        cfg = DreamConfig(vocab_size=15080, num_hidden_layers=1, mask_token_id=synth.MASK, pad_token_id=synth.END)
        sd = synth.make_weights(cfg, seed=0)
        sd["model.norm.weight"] = sd["model.norm.weight"] * peak
        m = DreamModel_length(cfg)
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).to(torch.bfloat16)
        st = m._hip_state(B, 46)
        st.hidden_rows.copy_(torch.randn(B * 46, 768, device=dev))
        st.rows_valid = True
        M = a.rows
        res = {}
        for fused in (True, False):
            st.fused_head = fused
            st.head_stats.zero_()
            philox = (99, 0, 1 << 20)
            st.head_and_sample(M, 0.1, 0.9, 0, _hip.CONF_NEG_ENTROPY, None, philox)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                conf, x0 = st.head_and_sample(M, 0.1, 0.9, 0, _hip.CONF_NEG_ENTROPY, None, philox)
            e1.record()
            torch.cuda.synchronize()
            res[fused] = (e0.elapsed_time(e1) / a.iters * 1e3, conf.clone(), x0.clone(), st.head_stats.cpu().tolist())
        same = torch.equal(res[True][1].view(torch.int32), res[False][1].view(torch.int32)) and torch.equal(res[True][2], res[False][2])
        rows, fb = res[True][3]
        print(f"peak x{peak:4.1f}  M={M}: fused {res[True][0]:7.1f} us | two-call {res[False][0]:7.1f} us | rows needing logits "
              f"{fb / max(rows, 1):6.1%} | mean top-1 confidence-mode value {float(res[False][1].mean()):.3f} | identical: {same}", flush=True)


if __name__ == "__main__":
    main()
