#!/usr/bin/env python3
"""Headline benchmark: denoised KG query sequences / second (BASELINE.json `metric`).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        # no launcher: starts the N ranks itself (fresh child processes, started
                                        # before this process touches the GPU) and relays rank 0's JSON line
    python bench.py --gpus 2 --rehearse # both ranks on cuda:0 over gloo: exercises the N > 1 code on a 1-GPU box

One bench "step" = one pass of the hot path over one batch: BASELINE config 2, the FB15k-237 mixed
deductive+abductive evaluation batch of 1024 sequences per GPU (512 abduction rows with 11 masks + 512 deduction
rows with 33 masks, as two `diffusion_generate` calls -- the reference never mixes layouts in one call,
custom_grpo2.py:593-599), 32 denoise steps each, temperature 0.1, top_p 0.9, alg 'entropy' (main.py:393-404),
bf16 weights/activations with fp32 accumulation, default model shape (10 x d768 x ff3072, V = 15080).
Every denoise step runs its full forward + head + sampling; nothing is cached or skipped (the categorical noise is
the same torch Philox stream the reference consumes, evaluated inside the sampler: --materialise-noise fills it with
torch instead, same results).
Inputs are synthetic (dark_amd.synth), resident in HBM before the timed region; weights are random-init.
Multi-GPU: rows are sharded data-parallel (weak scaling: 1024 sequences per GPU), no collective inside the loop;
the per-batch result gather (sequences, fixed-width records) over RCCL is inside the timed region.

The JSON line also carries
  roofline:     the dominant kernel (gate/up GEMM with fused SwiGLU, 50% of the forward FLOPs), timed live with
                HIP events on the launch stream (dark_prof_* hook of the C ABI) during the timed region;
  cpu_baseline: the CPU oracle (oracle/dream_oracle.py, a PyTorch-CPU fp32 restatement of the reference path;
                kind "port") on a bounded sample of the same workload, on this box's host cores (rank 0, N=1).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3    # fp32 matrix


def algorithmic_flops_per_call(cfg, B, N, steps, masks_per_row, executed=False):
    """F_body(N) = N*L*(8d^2 + 6*d*ff + 4*N*d) per sequence-step, F_head = 2*d*V per masked row.
    executed=True counts what this build actually runs: in the LAST layer o_proj and the MLP (2d^2 + 6*d*ff per
    row) are evaluated for the masked rows only (dark_forward_hidden_rows) -- the other rows' outputs are read by
    nothing in the reference either (generation_utils.py:501 drops them)."""
    from dark_amd.generation_utils import transfer_count
    d, ff, L, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.vocab_size
    body = N * L * (8 * d * d + 6 * d * ff + 4 * N * d)
    tail_row = 2 * d * d + 6 * d * ff
    ts = torch.linspace(1, 1e-3, steps + 1)
    left, head_rows = masks_per_row, 0
    for i in range(steps):
        head_rows += left
        left -= transfer_count(left, 1, ts[i], ts[i + 1], i == steps - 1)
    total = steps * body + head_rows * 2 * d * V
    if executed and L > 0:
        total -= steps * N * tail_row - head_rows * tail_row
    return B * total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="sequences per GPU per step (half abd, half ded)")
    ap.add_argument("--denoise-steps", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-peaked-leg", action="store_true", help="skip the value_peaked_rows leg (A/B runs)")
    ap.add_argument("--materialise-noise", action="store_true",
                    help="fill the [M,V] categorical noise with torch instead of computing it in the sampler")
    ap.add_argument("--cpu-sample", type=int, default=48, help="sequences in the CPU-oracle sample")
    ap.add_argument("--gemm-variant", type=int, default=0, choices=[0, 1, 2, 4, 5, 6, 7, 11, 12, 13, 14] + list(range(1000, 1128)),
                    help="A/B hook: 0 = the library's choice, 1 = 128-row GEMM kernels only, 2 / 4 = the persistent 256x256 "
                         "kernel (plain / staggered) wherever it supports the shape; 6 / 7 / 11 = the library's choice "
                         "without the down_proj routing / without the q/k/v routing / with o_proj on the persistent kernel")
    ap.add_argument("--sampler-single-stage", action="store_true",
                    help="A/B: the single-stage sampler kernel only (default: the two-stage form where it applies)")
    ap.add_argument("--residual-fp32", action="store_true",
                    help="A/B: carry the bf16 model's residual stream in fp32 (default: the model dtype, as the "
                         "reference's bf16 model does)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on ONE card: every rank uses cuda:0 and the collectives run over gloo (a functional "
                         "rehearsal of the multi-GPU path, not a measurement)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives at world size 1 too (takes the nccl = "
                         "RCCL branch on a 1-GPU box; needs RANK / WORLD_SIZE / MASTER_* as a launcher sets them)")
    ap.add_argument("--workload", default="denoise", choices=["denoise", "expand", "logprob"],
                    help="denoise = the headline (BASELINE config 2); expand / logprob = the widened paths of SURVEY "
                         "8f-1 / 8f-3 on one GPU (tools/bench_expand.py, tools/bench_logprob.py) with their CPU-oracle legs")
    args = ap.parse_args()
    if args.workload != "denoise":
        return widened(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)  # nothing has touched the GPU in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:   # (also under an external launcher, which does not set a per-rank thread count)
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 16) // world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    grouped = world > 1 or args.force_dist
    # ONE JSON line on stdout and nothing else: RCCL prints a version banner and gloo a connection note on file
    # descriptor 1 when a communicator comes up; until the result is printed, descriptor 1 points at stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if grouped:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from dark_amd import DreamConfig, DreamGenerationConfig, DreamModel_length, _hip, dispatch, synth

    dispatch.FORCE_COLLECTIVES = bool(args.force_dist)
    lib = _hip.load()
    _hip.check(lib.dark_dbg_set_gemm_variant(args.gemm_variant))
    _hip.check(lib.dark_dbg_set_sampler_stages(1 if args.sampler_single_stage else 0))
    cfg = DreamConfig(vocab_size=synth.vocab_size(**synth.FB15K237), mask_token_id=synth.MASK,
                      pad_token_id=synth.END)
    weights = synth.make_weights(cfg, seed=0)
    model = DreamModel_length(cfg)
    model.residual_in_fp32 = bool(args.residual_fp32)
    model.load_state_dict(weights, strict=True)
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = model.to(dev).to(tdt)

    half = args.batch // 2
    ids_abd = synth.make_inputs(half, "abd", seed=1 + 2 * rank).to(dev)
    ids_ded = synth.make_inputs(args.batch - half, "ded", seed=2 + 2 * rank).to(dev)
    gc = DreamGenerationConfig(mask_token_id=synth.MASK, pad_token_id=synth.END, bos_token_id=synth.BOS,
                               eos_token_id=synth.END, sep_token_id=synth.SEP, max_length=46)
    gen_kw = dict(steps=args.denoise_steps, temperature=0.1, alg="entropy", alg_temp=0, top_p=0.9,
                  number_transfer_tokens=1, skip_noop_steps=False,  # headline: every step runs its forward
                  lazy_noise=not args.materialise_noise)
    torch.manual_seed(2 + rank)

    import warnings
    warnings.filterwarnings("ignore", message="Padding was detected")

    def one_step(mdl=None, **over):
        seqs = []
        for ids in (ids_abd, ids_ded):
            seqs.append((mdl or model).diffusion_generate(ids, generation_config=gc, **dict(gen_kw, **over)))
        out = torch.cat(seqs, 0)
        if grouped:  # per-batch result gather (replaces all_gather_object of mdm_unify.py:436-438)
            out = dispatch.gather_rows(out, out.shape[0] * world)
        return out

    def fence():
        torch.cuda.synchronize(dev)
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = one_step()
    fence()
    EPI_SWIGLU = 2
    rows_full = half * 46  # the full-size gate/up launches (9 of 10 layers); the last layer's tail has fewer rows
    lib.dark_prof_select_gemm_rows(EPI_SWIGLU, rows_full)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    fence()
    dt = time.perf_counter() - t0
    ms, fl, n = C.c_double(), C.c_double(), C.c_int()
    _hip.check(lib.dark_prof_read(C.byref(ms), C.byref(fl), C.byref(n)))
    lib.dark_prof_select_gemm(-1)
    assert out.shape == (args.batch * world, 46) and int((out[:, 0] != synth.BOS).sum()) == 0

    # secondary figure: the library default (skip_noop_steps=True) reuses the hidden states of an unchanged x on
    # the steps whose transfer count is 0 (22 of 32 for the abduction half); same tokens, same RNG stream
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        one_step(skip_noop_steps=True)
    fence()
    dt_skip = time.perf_counter() - t1

    # third figure: the same workload on weights whose rows are as peaked as a trained checkpoint's (final-norm weight
    # x 6, as tools/head_fused_bench.py --peaks 6 makes them), so that the fused head + sampler (dark_head_sample_rows:
    # no [M, V] logits) stays on; on the random-init weights of the headline the library's own statistics switch it off
    # (most rows are flat and need their logits anyway).  Every step runs its forward, as in the headline.
    dt_peaked, peaked_stats = None, None
    if args.dtype == "bf16" and not args.no_peaked_leg:
        w6 = dict(weights)
        w6["model.norm.weight"] = weights["model.norm.weight"] * 6.0
        model6 = DreamModel_length(cfg)
        model6.residual_in_fp32 = bool(args.residual_fp32)
        model6.load_state_dict(w6, strict=True)
        model6 = model6.to(dev).to(tdt)
        one_step(model6)
        fence()
        for st_ in model6._states.values():
            st_.head_stats.zero_()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            one_step(model6)
        fence()
        dt_peaked = time.perf_counter() - t2
        hs = sum(st_.head_stats.cpu().long() for st_ in model6._states.values())
        peaked_stats = {"rows_sampled_by_fused_head": int(hs[0]), "rows_that_needed_logits": int(hs[1])}
        del model6

    if grouped:
        t = torch.tensor([dt, dt_skip, dt_peaked or 0.0], device="cpu" if args.rehearse else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_skip = float(t[0]), float(t[1])
        dt_peaked = float(t[2]) if dt_peaked is not None else None
    seq_per_s = args.batch * world * args.steps / dt

    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
    alg_flops = (algorithmic_flops_per_call(cfg, half, 46, args.denoise_steps, 11)
                 + algorithmic_flops_per_call(cfg, args.batch - half, 46, args.denoise_steps, 33))
    exe_flops = (algorithmic_flops_per_call(cfg, half, 46, args.denoise_steps, 11, True)
                 + algorithmic_flops_per_call(cfg, args.batch - half, 46, args.denoise_steps, 33, True))
    achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    result = {
        "metric": "denoised KG query sequences/sec (whole node) at T steps",
        "value": round(seq_per_s, 2), "unit": "sequences/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "BASELINE config 2: FB15k-237 mixed deductive+abductive eval, default DARK shape "
                               "(10L d768 ff3072 H8, V=15080, N=46)",
                   "batch_per_gpu": args.batch, "denoise_steps": args.denoise_steps, "temperature": 0.1,
                   "top_p": 0.9, "alg": "entropy", "parallelism": f"dp{world}", "weights": "random-init",
                   "residual_stream": "f32" if (args.residual_fp32 or args.dtype == "f32") else "bf16",
                   **({"collectives": "nccl (forced at world size 1)"} if args.force_dist and not args.rehearse else {})},
        "value_skip_noop_steps": round(args.batch * world * args.steps / dt_skip, 2),
        **({"value_peaked_rows": round(args.batch * world * args.steps / dt_peaked, 2),
            "peaked_rows": dict(peaked_stats, weights="random-init, final-norm weight x 6 (rows as peaked as a trained "
                                                     "checkpoint's at T = 0.1); head + sampler fused, no [M, V] logits")}
           if dt_peaked else {}),
        # SURVEY 8(d)'s algorithmic FLOPs (every row through every layer, head on masked rows) over the dense peak ...
        "forward_mfma_frac": round(alg_flops * args.steps * world / dt / 1e12 / (peak * world), 4),
        # ... and the FLOPs this build executes (last layer's o_proj / MLP on the masked rows only)
        "forward_mfma_frac_executed": round(exe_flops * args.steps * world / dt / 1e12 / (peak * world), 4),
        "roofline": {"bound": "mfma", "kernel": (({0: "gemm_bf16_s256<2, 0, 10, 10>", 14: "gemm_bf16_s256<2, 0, 10, 0>", 4: "gemm_bf16_s256<2, 0, 10, 10>", 2: "gemm_bf16_p256<2, false>"}.get(
                                    args.gemm_variant, "gemm_bf16_kernel16<2, 192>") if args.dtype == "bf16" else "gemm_f32_kernel<2>")
                                + f" = EPI_SWIGLU (gate/up projection + SiLU*up, the {rows_full}-row launches of layers 0..L-2, "
                                  "N=6144, K=768)"),
                     "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4),
                     # HBM-side bytes per launch from rocprofv3 PMC (FETCH_SIZE doubled per the gfx950 note, +
                     # WRITE_SIZE), separate passes, profiles/r02zzf_pmc.txt; valid for the default shape and kernel
                     # (rocprofv3 cannot run inside this process: the sidecar profiles/roofline_traffic.json holds the
                     # per-launch figure of the PMC passes over THIS command on the build it names, written by
                     # tools/profile_round.sh; null when it is absent or was taken for another shape / kernel)
                     **roofline_traffic(args),
                     "algorithmic_bytes": 2.0 * (half * 46 * 768 + 6144 * 768 + half * 46 * 3072),
                     "launches": n.value,   # bracketed launches (the hook brackets at most 8192 per read: all of a default run)
                     "launches_cap": 8192,
                     "avg_launch_ms": round(ms.value / max(n.value, 1), 4)},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg, weights, args, synth)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if grouped:
        os.dup2(2, 1)   # (the teardown may talk as well)
        dist.destroy_process_group()


def roofline_traffic(args):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE doubled per the gfx950
    note of the microarchitecture guide + WRITE_SIZE, separate passes)."""
    path = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        with open(path) as f:
            side = json.load(f)
    except (OSError, ValueError):
        return {"traffic": None, "traffic_source": None}
    ok = (args.dtype == side.get("dtype") and args.batch == side.get("batch") and args.gemm_variant == 0
          and not args.residual_fp32)
    return {"traffic": float(side["traffic_bytes_per_launch"]) if ok else None,
            "traffic_source": side.get("source") if ok else None}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* as torch.distributed.run would set them), wait for all, relay rank 0's stdout.  The parent never
    initialises the GPU and never replaces itself: the ranks are children and its exit code is theirs."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    # host threads per rank: the per-step sequencing is one Python thread; torch's intra-op pool must not oversubscribe
    # the host when N ranks share it (the box sets OMP_NUM_THREADS for ONE process)
    box_threads = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or min(16, os.cpu_count() or 16)
    per_rank = str(max(1, box_threads // args.gpus))
    for r in range(args.gpus):
        env = dict(os.environ, OMP_NUM_THREADS=per_rank, MKL_NUM_THREADS=per_rank, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # a rank that dies would leave the others waiting in a collective: watch them and stop the rest (by PID)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    reader.join(timeout=5)
    out = "".join(c for c in chunks if c)
    sys.stdout.write(out)
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"rank exit codes {codes}")


def widened(args):
    """Single-GPU figures of the widened paths; the CPU legs run here because only bench.py may time oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from dark_amd import synth
    if args.workload == "expand":
        import bench_expand as T
        res, w, cfg, prompts = T.run(T.parse([]))
        if not args.no_cpu_baseline:
            from oracle import dream_oracle as O  # baseline leg only
            ocfg = O.OracleConfig(**{k: getattr(cfg, k) for k in O.OracleConfig.__dataclass_fields__})
            n = 4
            torch.manual_seed(0)
            t0 = time.perf_counter()
            for p in prompts[:n]:
                O.expand_loop(w, ocfg, p, max_length=50, mask_token_id=synth.MASK, eos_token_id=synth.END,
                              delete_token_id=synth.END, expand_token_id=synth.EXPAND, temperature=0.1, top_p=0.9,
                              alg="entropy", alg_temp=0)
            res["cpu_baseline"] = {"value": round(n / (time.perf_counter() - t0), 3), "unit": "sequences/s",
                                   "cores": torch.get_num_threads(), "kind": "port", "sample": f"{n} prompts"}
    else:
        import bench_logprob as T
        res, w, cfg, ids, target = T.run(T.parse([]))
        if not args.no_cpu_baseline:
            from oracle import dream_oracle as O  # baseline leg only
            from oracle import grpo_oracle as G
            ocfg = O.OracleConfig(**{k: getattr(cfg, k) for k in O.OracleConfig.__dataclass_fields__})
            n = 8
            t0 = time.perf_counter()
            G.per_token_logps(w, ocfg, ids[:1, :n], target[:1, :n].long(), [1], synth.MASK)
            res["cpu_baseline"] = {"value": round(n / (time.perf_counter() - t0), 2), "unit": "completions/s",
                                   "cores": torch.get_num_threads(), "kind": "port", "sample": f"{n} completions"}
    print(json.dumps(res), flush=True)


def cpu_baseline(cfg, weights, args, synth):
    """The oracle on a bounded sample of the same workload: `cpu_sample` sequences (half abd, half ded), all
    denoise steps, fp32, torch's intra-op threads as configured on this box."""
    from oracle import dream_oracle as O  # baseline leg only; never on the product path
    ocfg = O.OracleConfig(**{k: getattr(cfg, k) for k in O.OracleConfig.__dataclass_fields__})
    n = max(2, args.cpu_sample)
    threads = torch.get_num_threads()
    torch.manual_seed(2)
    t0 = time.perf_counter()
    for kind, rows in (("abd", n // 2), ("ded", n - n // 2)):
        ids = synth.make_inputs(rows, kind, seed=1)
        O.sample_loop(weights, ocfg, ids, max_length=46, steps=args.denoise_steps, mask_token_id=synth.MASK,
                      alg="entropy", alg_temp=0, temperature=0.1, top_p=0.9)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "sequences/s", "cores": threads, "kind": "port",
            "sample": f"{n} sequences ({n // 2} abd + {n - n // 2} ded) x {args.denoise_steps} denoise steps, fp32 "
                      f"PyTorch-CPU oracle, {dt:.1f} s"}


if __name__ == "__main__":
    main()
