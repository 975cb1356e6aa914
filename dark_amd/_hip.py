"""ctypes binding of ``libdark_hip.so`` (C ABI in ``include/dark_hip.h``).

The library is the product: if it is missing or cannot be loaded this module raises, it never
falls back to a CPU or PyTorch implementation.  Build it with ``python -c "import __graft_entry__ as g;
g.build()"`` or ``make -C dark_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DARK_HIP_LIB: measurement tools point this at the probe build (libdark_hip_probes.so, `make -C dark_amd/csrc probes`);
# it must export the same ABI -- load() checks every symbol either way.
LIB_PATH = os.environ.get("DARK_HIP_LIB") or os.path.join(_HERE, "libdark_hip.so")

DARK_OK, DARK_ERR_ARG, DARK_ERR_HIP, DARK_ERR_UNSUPPORTED = 0, 1, 2, 3
DARK_F32, DARK_BF16 = 0, 1
CONF_PROB, CONF_MARGIN, CONF_NEG_ENTROPY = 0, 1, 2
ABI_VERSION = 7
SHARD_PART_BYTES = 280
SAMPLE_LDS_MAX_V = 32768   # DARK_SAMPLE_LDS_MAX_V: larger vocabularies sample from a registered row workspace


class LayerWeights(C.Structure):
    _fields_ = [("w_qkv", C.c_void_p), ("b_qkv", C.c_void_p), ("w_o", C.c_void_p), ("w_gu", C.c_void_p),
                ("w_down", C.c_void_p), ("ln1", C.c_void_p), ("ln2", C.c_void_p)]


class ModelDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("vocab", C.c_int32), ("vocab_pad", C.c_int32), ("d", C.c_int32),
                ("ff", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("head_dim", C.c_int32),
                ("max_pos", C.c_int32), ("rms_eps", C.c_float), ("resid_dtype", C.c_int32), ("embed", C.c_void_p),
                ("final_norm", C.c_void_p), ("lm_head", C.c_void_p), ("rope_cos", C.c_void_p),
                ("rope_sin", C.c_void_p), ("layers", C.POINTER(LayerWeights))]


class Gpt2LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_qkv", "b_qkv", "w_o", "b_o", "w_fc", "b_fc", "w_proj", "b_proj",
                                          "ln1_w", "ln1_b", "ln2_w", "ln2_b")]


class Gpt2Desc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("vocab", C.c_int32), ("vocab_pad", C.c_int32), ("d", C.c_int32),
                ("ff", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("head_dim", C.c_int32),
                ("n_positions", C.c_int32), ("ln_eps", C.c_float), ("resid_dtype", C.c_int32), ("wte", C.c_void_p),
                ("wpe", C.c_void_p),
                ("lnf_w", C.c_void_p), ("lnf_b", C.c_void_p), ("ident_cos", C.c_void_p), ("ident_sin", C.c_void_p),
                ("layers", C.POINTER(Gpt2LayerWeights))]


# name -> (restype, argtypes); every symbol include/dark_hip.h declares
SIGNATURES = {
    "dark_hip_abi_version": (C.c_int, []),
    "dark_hip_last_error": (C.c_char_p, []),
    "dark_forward_workspace_bytes": (C.c_size_t, [C.POINTER(ModelDesc), C.c_int, C.c_int]),
    "dark_head_workspace_bytes": (C.c_size_t, [C.POINTER(ModelDesc), C.c_int]),
    "dark_step_workspace_bytes": (C.c_size_t, [C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int]),
    "dark_step_workspace_offsets": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int,
                                              C.POINTER(C.c_size_t)]),
    "dark_forward_hidden": (C.c_int, [C.POINTER(ModelDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "dark_forward_hidden_begin": (C.c_int, [C.POINTER(ModelDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "dark_forward_hidden_rows": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dark_lm_head_rows": (C.c_int, [C.POINTER(ModelDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_int64, C.c_void_p]),
    "dark_noalloc_init": (C.c_int, []),
    "dark_noalloc_max_rows": (C.c_size_t, [C.POINTER(ModelDesc)]),
    "dark_lm_head_rows_noalloc": (C.c_int, [C.POINTER(ModelDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_int64, C.c_void_p]),
    "dark_mask_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "dark_sample_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dark_sample_rows_philox": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                          C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "dark_sampler_set_row_workspace": (C.c_int, [C.c_void_p, C.c_size_t]),
    "dark_philox_exponential": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "dark_final_norm_rows": (C.c_int, [C.POINTER(ModelDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "dark_lm_head_shard": (C.c_int, [C.POINTER(ModelDesc), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_int64, C.c_void_p]),
    "dark_shard_row_max": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "dark_shard_partials": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "dark_sample_from_shards": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64,
                                          C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dark_sample_rows_indexed": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64,
                                           C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dark_commit_topk": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_void_p]),
    "dark_commit_sampled": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "dark_scatter_tokens": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_void_p]),
    "dark_denoise_step": (C.c_int, [C.POINTER(ModelDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "dark_expand_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "dark_expand_block_logit": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dark_expand_commit_edit": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                          C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dark_expand_commit_edit_sampled": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                  C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                                  C.c_void_p, C.c_void_p]),
    "dark_logprob_workspace_bytes": (C.c_size_t, [C.POINTER(ModelDesc), C.c_int]),
    "dark_lm_head_logprob": (C.c_int, [C.POINTER(ModelDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "dark_logprob_from_logits": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p]),
    "dark_grpo_perturb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "dark_shift_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dark_grpo_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p]),
    "dark_build_layout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                    C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dark_split_strip": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dark_prof_select_gemm": (C.c_int, [C.c_int]),
    "dark_prof_select_gemm_rows": (C.c_int, [C.c_int, C.c_int]),
    "dark_prof_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "dark_dbg_gemm": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                C.c_int, C.c_int64, C.c_void_p]),
    "dark_dbg_set_gemm_variant": (C.c_int, [C.c_int]),
    "dark_dbg_set_sampler_stages": (C.c_int, [C.c_int]),
    "dark_dbg_attention": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dark_head_sample_workspace_bytes": (C.c_size_t, [C.POINTER(ModelDesc), C.c_int]),
    "dark_head_sample_rows": (C.c_int, [C.POINTER(ModelDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p,
                                        C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "dark_dbg_gemm_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                     C.c_void_p]),
    "dark_dbg_gemm_resid_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_float, C.c_void_p]),
    "dark_dbg_gemm_resid_norm_h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_float, C.c_void_p]),
    "dark_dbg_rmsnorm": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_float, C.c_void_p]),
    "dark_gpt2_forward_workspace_bytes": (C.c_size_t, [C.POINTER(Gpt2Desc), C.c_int, C.c_int]),
    "dark_gpt2_forward_hidden": (C.c_int, [C.POINTER(Gpt2Desc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "dark_gpt2_lm_head_rows": (C.c_int, [C.POINTER(Gpt2Desc), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_int64, C.c_void_p]),
    "dark_dbg_layernorm": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "dark_dbg_embed": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


class DarkHipError(RuntimeError):
    """A C-ABI call returned a DARK_ERR_* code (message from dark_hip_last_error)."""


def load() -> C.CDLL:
    """dlopen the library and bind every symbol; raises if anything is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library is the only implementation of this path. "
            "Build it with `make -C dark_amd/csrc` (hipcc, --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.dark_hip_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libdark_hip.so ABI version {got}, binding expects {ABI_VERSION}")
    _lib = lib
    # every entry point that samples registers the large-vocabulary row workspace first (a no-op for V <= 32768)
    for name, (m_at, v_at) in _SAMPLING_CALLS.items():
        setattr(lib, name, _with_row_workspace(getattr(lib, name), m_at, v_at))
    return lib


# (index of M, index of V or None = args[0] is a model descriptor) in the argument lists of the sampling entry points
_SAMPLING_CALLS = {"dark_sample_rows": (2, 3), "dark_sample_rows_philox": (2, 3), "dark_sample_rows_indexed": (2, 3),
                   "dark_head_sample_rows": (3, None), "dark_denoise_step": (6, None)}
_row_ws = {}   # device index -> uint8 tensor registered with dark_sampler_set_row_workspace


def ensure_sampler_rows(M: int, V: int) -> None:
    """Vocabularies beyond DARK_SAMPLE_LDS_MAX_V (WN18RR: 41 066): the sampler keeps its rows in a device workspace
    that the caller registers, because no entry point allocates.  One buffer per device (the CURRENT device, which
    the library requires to be the tensors' device), grown when a call needs more, shared by every model on it:
    sampler calls with V > DARK_SAMPLE_LDS_MAX_V that OVERLAP on one GPU (two streams, two host threads) race on it --
    the one-stream-at-a-time contract of include/dark_hip.h (dark_sampler_set_row_workspace)."""
    if V <= SAMPLE_LDS_MAX_V or M <= 0:
        return
    need = M * ((V + 3) & ~3) * 4
    dev = torch.cuda.current_device()
    buf = _row_ws.get(dev)
    if buf is None or buf.numel() < need:
        if buf is not None:
            # a sampler launch on ANOTHER stream may still read the old buffer: torch's allocator orders reuse on the
            # allocation stream only, so wait for the device before the old buffer is dropped (growth is rare)
            torch.cuda.synchronize(dev)
        buf = torch.empty(need, dtype=torch.uint8, device=torch.device("cuda", dev))
        check(_lib.dark_sampler_set_row_workspace(buf.data_ptr(), buf.numel()))
        _row_ws[dev] = buf


def _with_row_workspace(fn, m_at: int, v_at):
    import functools

    @functools.wraps(fn)
    def call(*args):
        V = args[v_at] if v_at is not None else getattr(getattr(args[0], "_obj", None), "vocab", 0)
        if V > SAMPLE_LDS_MAX_V:
            ensure_sampler_rows(int(args[m_at]), int(V))
        return fn(*args)
    return call


def check(rc: int) -> None:
    if rc != DARK_OK:
        msg = load().dark_hip_last_error()
        raise DarkHipError(f"dark_hip error {rc}: {msg.decode() if msg else ''}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous CUDA/HIP tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("dark_hip entry points take device memory; got a CPU tensor")
    if not t.is_contiguous():
        raise ValueError("dark_hip entry points take contiguous tensors")
    return t.data_ptr()


def device_guard(device_of):
    """Decorator: run the function with `device_of(*args, **kwargs)` as the current HIP device.  The library
    enqueues on the stream it is given and keeps per-device kernel attributes, so the tensors' device must be the
    current one (dark_hip refuses a stream of another device)."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*args, **kwargs):
            dev = torch.device(device_of(*args, **kwargs))
            if dev.type != "cuda":
                return fn(*args, **kwargs)
            with torch.cuda.device(dev):
                return fn(*args, **kwargs)
        return wrapped
    return deco


def stream_ptr(device: torch.device) -> int:
    """hipStream_t of torch's current stream on `device`, as an integer handle."""
    return torch.cuda.current_stream(device).cuda_stream
