"""The multi-GPU dispatcher with the REAL model (round 1 only ran it with a fake one): two rank processes share
cuda:0 and talk over gloo (tests/_dispatch_worker.py), the parent compares with unsharded calls.  Covers the
`exact_ragged_k` all-reduce of the mask count (`_count_reduce(st.count[:1], B)`), per-shard RNG parity, the record
gather, and `bench.py --gpus 2` starting its own ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, prefix, mode="dp"):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dispatch_worker.py"), prefix, mode], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    return [torch.load(f"{prefix}.rank{r}.pt") for r in range(world)]


def test_generate_sharded_real_model_world2(tmp_path):
    import _dispatch_worker as W
    from dark_amd import dispatch
    res = _run_ranks(2, str(tmp_path / "out"))
    for key in res[0]:
        assert torch.equal(res[0][key], res[1][key]), f"{key}: ranks disagree after the gather"
    m = W.build()
    kw = dict(steps=8, temperature=0.0, alg="entropy", alg_temp=0, top_p=0.9)
    rag = W.ragged_inputs().to("cuda:0")
    whole = m.diffusion_generate(rag, generation_config=W.gen_config(), **kw).cpu()
    # ragged masks: only the all-reduced mask count reproduces the unsharded k-schedule ...
    assert torch.equal(res[0]["ragged_exact"], whole)
    # ... and the shard-local count does not (rows 0-5 hold 50 masks, rows 6-11 hold 66: different averages)
    halves = torch.cat([m.diffusion_generate(rag[lo:hi], generation_config=W.gen_config(), **kw).cpu()
                        for lo, hi in (dispatch.shard_bounds(12, 2, 0), dispatch.shard_bounds(12, 2, 1))])
    assert torch.equal(res[0]["ragged_local_k"], halves)
    assert not torch.equal(halves, whole), "the fixture must make the two schedules differ"
    # stochastic parity is per shard: same seed + same rows => the unsharded call on those rows alone
    uni = __import__("dark_amd").synth.make_inputs(10, "ded", seed=5).to("cuda:0")
    parts = []
    for r in range(2):
        lo, hi = dispatch.shard_bounds(10, 2, r)
        torch.manual_seed(100 + r)
        parts.append(m.diffusion_generate(uni[lo:hi], generation_config=W.gen_config(), steps=8, temperature=0.5,
                                          alg="entropy", alg_temp=0, top_p=0.9).cpu())
    assert torch.equal(res[0]["uniform_t05"], torch.cat(parts))
    assert res[0]["scores"].tolist() == [[0, 1], [2, 3], [4, 5], [10, 11], [12, 13], [14, 15]]
    assert res[0]["pattern_id"].tolist() == [1, 2, 3, 11, 12, 13]


def test_rccl_single_rank_collectives(tmp_path):
    """The `nccl` (= RCCL) branches of dispatch / vocab_shard, which two ranks sharing one card cannot take (RCCL refuses
    a device twice): one rank, every collective forced to run, results equal to the plain calls."""
    res = _run_ranks(1, str(tmp_path / "nccl"), "nccl1")[0]
    assert torch.equal(res["nccl_ragged"], res["plain_ragged"])
    assert torch.equal(res["nccl_shard_head"], res["plain_t05"])
    assert torch.equal(res["nccl_bf16_shard_head"], res["plain_bf16"])
    assert res["scores"].tolist() == [[0, 1], [2, 3], [4, 5]] and res["pattern_id"].tolist() == [1, 2, 3]
    assert res["f64_max"].tolist() == [1.5, 2.5]


def test_bench_nccl_branch_single_rank():
    """bench.py's `nccl` branch (process-group init on the device, barrier, gather of the sequences, max over ranks of
    the timings) at world size 1 under the launcher's environment."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "1",
                        "--warmup", "0", "--batch", "64", "--denoise-steps", "8", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["config"]["collectives"] == "nccl (forced at world size 1)"


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher must start two ranks itself and print ONE JSON line (rc 0);
    --rehearse puts both on cuda:0 over gloo (a functional rehearsal: the value is not a measurement)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "1",
                        "--warmup", "0", "--batch", "64", "--denoise-steps", "8", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["value"] > 0
