"""Data-parallel training end to end (ref:config/trainer/ddp.yaml:5, ref:icv_src/icv_module.py:160-209): two
fresh rank processes (torch.distributed.run, gloo so both can share this box's one GPU; RCCL on a real node) each run
ICVTrainer.micro_batch on their shard of every accumulation window — teacher forward, student forward/backward, ONE
all-reduce of [alpha.grad | icv.grad | kl], clipped fused AdamW.  Asserted:
  * both ranks end with BIT-IDENTICAL parameters (same reduced gradient, same update);
  * they equal, to fp32 summation-order noise, the single-process run that accumulates the union of the micro-batches itself;
  * the parameters moved, and the logged kl is the mean over all micro-batches of the window.
"""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_trainer_matches_single_process_union_batch(tmp_path):
    steps, accumulate = 2, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(HERE / "dp_worker.py"), "--out", str(tmp_path), "--steps", str(steps),
           "--accumulate", str(accumulate), "--backend", "gloo"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r0, r1 = (torch.load(tmp_path / f"rank{i}.pt") for i in range(2))
    for name in ("alpha", "icv"):
        assert torch.equal(r0[name], r1[name]), f"{name}: ranks diverged"
    sys.path.insert(0, str(HERE))
    import dp_worker
    one = dp_worker.run(0, 1, steps, 2 * accumulate)                  # the same 4 micro-batches per window, one process
    torch.manual_seed(426)
    from icv_src.icv_encoder.global_icv_encoder import GlobalICVEncoder
    from licv.config import IDEFICS_TINY
    init = GlobalICVEncoder(IDEFICS_TINY.hidden_size, IDEFICS_TINY.num_layers, alpha_init_value=0.3, use_sigmoid=True)
    for name in ("alpha", "icv"):
        p0 = getattr(init, name).detach()
        d_dp, d_one = r0[name] - p0, one[name] - p0
        assert float(d_one.abs().max()) > 0, f"{name} did not move"
        # AdamW normalises the step, so compare the updates themselves: identical data, differently ordered fp32 sums and bf16
        # kernels that see B = 2 per micro-batch on both sides
        # (Adam's first steps are ~lr*sign(g): an element whose gradient sits at the summation-order noise floor may flip sign,
        # hence a relative-L2 bar plus "at most 1 % of the elements off by more than 2 % of the largest update")
        rel = float((d_dp - d_one).norm() / d_one.norm())
        off = float(((d_dp - d_one).abs() > 2e-2 * d_one.abs().max()).float().mean())
        assert rel <= 2e-2 and off <= 0.01, f"{name}: DP update vs union-batch update: relative L2 {rel:.3e}, {100 * off:.2f} % elements off"
    for a, b in zip(r0["logs"], one["logs"]):
        assert abs(a["kl_loss"] - b["kl_loss"]) <= 1e-3 * abs(b["kl_loss"]) + 1e-6       # mean of rank means == union mean
        assert abs(a["grad_norm"] - b["grad_norm"]) <= 2e-2 * b["grad_norm"] + 1e-8
    assert r0["logs"][0]["kl_loss"] == r1["logs"][0]["kl_loss"]


@pytest.mark.parametrize("workload", ["idefics_mid_debug", "idefics_mid_train_debug"])
def test_bench_two_ranks_reports_what_the_process_group_saw(workload):
    """`python bench.py --gpus 2` starts its own two ranks (a child torch.distributed.run, before the parent touches the GPU) and its
    JSON line proves what ran: n_gpus, the backend the process group reports, the world size IT saw and one device name per rank;
    the training workload also times its one collective.  gloo here (both ranks share this box's one GPU); RCCL on a real node."""
    import json
    root = HERE.parent
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "0",
           "--no-cpu-baseline", "--no-gpu-baseline", "--workload", workload]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2 and d["dist_backend"] == "gloo"
    assert len(d["devices"]) == 2 and d["devices"][0].startswith("rank0:") and d["devices"][1].startswith("rank1:")
    assert d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak" and d["value"] > 0
    if "train" in workload:
        ar = d["allreduce"]
        assert ar["optimizer_steps_timed"] >= 1 and ar["per_optimizer_step_ms_median"] > 0 and ar["bytes"] > 0
