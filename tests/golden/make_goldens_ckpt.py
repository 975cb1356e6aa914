"""Write a checkpoint directory with the REFERENCE's own writer (build container only).

    python tests/golden/make_goldens_ckpt.py

The producer of the checkpoints this library has to read is HF ``Trainer.save_model(<out>/final)``
(akgr/abduction_model/main.py:225-226), i.e. ``PreTrainedModel.save_pretrained`` of the reference's
``DreamModel_length`` followed by ``DreamGenerationConfig`` being re-read from the same directory
(modeling_dream_length.py:549-600).  This script imports the reference (same two harness shims as
make_goldens.py), builds a tiny model with seeded weights and lets the reference write

    tests/golden/ref_ckpt_tiny/          config.json, model.safetensors, generation_config.json
    tests/golden/ref_ckpt_tiny_sharded/  the same weights cut into shards + model.safetensors.index.json

plus ``reference_goldens_ckpt.npz``: input ids and the logits the reference computes AFTER reloading the
directory with its own ``from_pretrained``.  The directories hold data written by library code (json + safetensors),
nothing of the reference's source.
"""
import json
import os
import shutil
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/akgr/abduction_model/model")

import numpy as np
import torch
from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS


def _default_rope(config=None, device=None, seq_len=None, **kw):
    hd = config.hidden_size // config.num_attention_heads
    inv = 1.0 / (config.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    return inv, 1.0


ROPE_INIT_FUNCTIONS.setdefault("default", _default_rope)

from custom_mydream.configuration_dream import DreamConfig  # noqa: E402
from custom_mydream import generation_utils as ref_gen  # noqa: E402
from custom_mydream.modeling_dream_length import DreamModel_length  # noqa: E402

ref_gen.DreamGenerationConfig.validate = lambda self, *a, **k: None
# shim 3 (this script only): transformers 5.x wants `_tied_weights_keys` as a dict; the reference declares the 4.50 list
# form (modeling_dream_length.py:737).  Nothing is tied (tie_word_embeddings=False, configuration_dream.py:42).
DreamModel_length._tied_weights_keys = {}

from dark_amd import synth  # noqa: E402

TINY = dict(vocab_size=256, hidden_size=192, intermediate_size=256, num_hidden_layers=1,
            num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=64,
            mask_token_id=synth.MASK, pad_token_id=synth.END)


def main():
    torch.manual_seed(0)
    cfg = DreamConfig(**TINY)
    model = DreamModel_length(cfg).eval()
    w = synth.make_weights(cfg, seed=11, rich=True)
    model.load_state_dict(w, strict=True)
    # the generation settings main.test_batch builds (main.py:355-363) travel with the checkpoint
    model.generation_config = ref_gen.DreamGenerationConfig(
        mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS, eos_token_id=synth.END,
        sep_token_id=synth.SEP, expand_token_id=synth.EXPAND, delete_token_id=synth.END, max_length=46, steps=16,
        temperature=0.1, top_p=0.9, alg="entropy", alg_temp=0.0)
    meta = {"torch": torch.__version__, "tiny": TINY, "files": {}}
    for name, kw in (("ref_ckpt_tiny", {}), ("ref_ckpt_tiny_sharded", {"max_shard_size": "300KB"})):
        d = os.path.join(HERE, name)
        shutil.rmtree(d, ignore_errors=True)
        model.save_pretrained(d, **kw)
        if not os.path.exists(os.path.join(d, "generation_config.json")):
            model.generation_config.save_pretrained(d)
        meta["files"][name] = sorted(os.listdir(d))
    # what the reference computes after reading its own directory back
    ids = torch.cat([synth.make_inputs(2, "abd", nentity=120, nrelation=35, seed=3), torch.full((2, 1), synth.MASK)], 1)  # 100 + 120 + 35 + 1 = 256
    logits = {}
    for name in ("ref_ckpt_tiny", "ref_ckpt_tiny_sharded"):
        m2 = DreamModel_length.from_pretrained(os.path.join(HERE, name)).eval()
        # shim 4 (this script only): transformers 5.x builds the model on the meta device and leaves the reference's
        # non-persistent `inv_freq` buffer (modeling_dream_length.py:123) uninitialised after from_pretrained (observed:
        # garbage values, different on every load); the pinned 4.50.2 computes it in __init__.  Recompute it as __init__ does.
        inv, _ = _default_rope(m2.config)
        m2.model.rotary_emb.inv_freq = inv
        m2.model.rotary_emb.original_inv_freq = inv
        sd2 = m2.state_dict()
        assert all(torch.equal(sd2[k], v) for k, v in model.state_dict().items()), "reloaded weights differ"
        with torch.no_grad():
            logits[name] = m2(ids).logits.float().numpy()
        gc = m2.generation_config
        meta[name + "_generation"] = {k: getattr(gc, k) for k in ("steps", "temperature", "top_p", "alg", "alg_temp",
                                                                    "mask_token_id", "max_length", "sep_token_id")}
    assert np.array_equal(logits["ref_ckpt_tiny"], logits["ref_ckpt_tiny_sharded"])
    with torch.no_grad():
        direct = model(ids).logits.float().numpy()
    assert np.array_equal(direct, logits["ref_ckpt_tiny"]), "the reference's round trip changed its own logits"
    np.savez_compressed(os.path.join(HERE, "reference_goldens_ckpt.npz"), ids=ids.numpy(),
                        logits=logits["ref_ckpt_tiny"])
    with open(os.path.join(HERE, "reference_goldens_ckpt.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True, default=str)
    print(json.dumps(meta["files"], indent=1))


if __name__ == "__main__":
    main()
