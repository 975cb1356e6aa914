"""bf16 parity against the REFERENCE's own bf16 model (tests/golden/make_goldens_bf16.py: `model.to(torch.bfloat16)` run
on the CPU at the full 10-layer default shape; logits of the 88 rows the sampler reads).

bf16 results cannot be bit-equal across implementations (every GEMM sums in its own order and rounds to 8 bits), so the
bar is stated relative to what the reference itself does: its bf16 logits differ from its fp32 logits by
`ref_bf16_rms_err` / `ref_bf16_max_err` (in the fixture's json).  This library's bf16 model must be (1) as close to the
fp32 logits as the reference's bf16 model is (rms error <= 1.1 x, max error <= 1.25 x the reference's own; measured on
MI355X: rms 0.0058 with the bf16 residual stream, 0.0047 with the fp32 stream, against the reference's 0.0062, on a logit
range of 4.7), in both residual-stream modes, and (2) at least as close to the reference's bf16 logits as two
independent bf16 evaluations of the same network would be (rms of the difference <= the two errors added in
quadrature; measured 0.0056 against 0.0085: the shared rounding points correlate the errors)."""
import json
import os

import numpy as np
import pytest
import torch

from dark_amd import DreamConfig, DreamModel_length, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
HERE = os.path.dirname(os.path.abspath(__file__))


def _model(cfg_kw, dtype, residual_in_fp32=False):
    cfg = DreamConfig(**cfg_kw)
    m = DreamModel_length(cfg)
    m.residual_in_fp32 = residual_in_fp32
    m.load_state_dict(synth.make_weights(cfg, seed=0, rich=False), strict=True)
    return m.to(DEV).to(dtype)


def test_bf16_logits_against_the_references_bf16_model():
    meta = json.load(open(os.path.join(HERE, "golden", "reference_goldens_bf16.json")))
    g = np.load(os.path.join(HERE, "golden", "reference_goldens_bf16.npz"))
    ref_stats = next(c for c in meta["cases"] if c["layers"] == 10)
    ids = torch.from_numpy(g["ids"]).to(DEV)
    src = torch.from_numpy(g["src_rows"]).long()
    ref16 = torch.from_numpy(g["L10_bf16_bits"].view(np.int16)).view(torch.bfloat16).float()
    V = meta["real"]["vocab_size"]

    def rows(m):
        return m(ids).logits.float().reshape(-1, V).cpu()[src]

    f32 = rows(_model(meta["real"], torch.float32))
    rng = float(f32.max() - f32.min())
    assert abs(rng - ref_stats["logit_range"]) < 1e-3 * rng          # same fp32 network as the fixture's
    # the reference's bf16 model against this fp32 model reproduces the fixture's own statistics
    ref_rms = float((ref16 - f32).pow(2).mean().sqrt())
    ref_max = float((ref16 - f32).abs().max())
    assert abs(ref_rms - ref_stats["ref_bf16_rms_err"]) < 0.02 * ref_rms and abs(ref_max - ref_stats["ref_bf16_max_err"]) < 1e-3
    for fp32_stream in (False, True):
        got = rows(_model(meta["real"], torch.bfloat16, fp32_stream))
        rms = float((got - f32).pow(2).mean().sqrt())
        mx = float((got - f32).abs().max())
        cross = float((got - ref16).pow(2).mean().sqrt())
        agree = float((got.argmax(-1) == f32.argmax(-1)).float().mean())
        print(f"bf16 vs fp32 logits, 10 layers, 88 rows x {V} (range {rng:.2f}): "
              f"{'fp32' if fp32_stream else 'bf16'} stream rms {rms:.5f} max {mx:.5f} arg-max agreement {agree:.3f}; "
              f"reference's bf16 model rms {ref_rms:.5f} max {ref_max:.5f}; rms(this - reference bf16) {cross:.5f}")
        assert rms <= 1.1 * ref_rms, (fp32_stream, rms, ref_rms)        # (1) no further from fp32 than the reference's bf16
        assert mx <= 1.25 * ref_max, (fp32_stream, mx, ref_max)
        assert cross <= (rms ** 2 + ref_rms ** 2) ** 0.5, (fp32_stream, cross)          # (2)
        assert agree >= ref_stats["ref_bf16_argmax_agree"] - 0.03
