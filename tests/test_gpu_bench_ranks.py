"""bench.py --gpus 2 on ONE GPU: the ranks are started by bench.py itself (a torch.distributed.run child), share the
GPU under PBSO_BENCH_BACKEND=gloo, and the line's headline is the configuration AS WRITTEN (--objects in total, sharded:
BASELINE configs[3]) with the gather, its cost legs and the weak (--objects per GPU) side leg.  The data path is the one
N GPUs run; only the collective's transport differs (gloo through the host stands in for the device group's RCCL calls,
which need one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_print_one_line_with_gather_and_strong_leg():
    env = dict(os.environ, PBSO_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--objects", "96", "--buffers", "86", "--steps", "3", "--warmup", "1",
                        "--settle", "2", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                     # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["gather"] is True and d["config"]["group_ranks"] == 2
    assert d["config"]["objects_total"] == 96                # --objects is the JOB's object count (BASELINE configs[3] as written)
    # the transport is named for what it is: gloo here, never "RCCL"
    assert d["config"]["rccl_ranks"] is None and d["config"]["backend"] == "gloo" and "gloo" in d["config"]["workload"] and "RCCL all-gather" not in d["config"]["workload"]
    assert d["dtype"] == "f32" and d["config"]["recurrence_form"] == "block"
    assert d["config"]["launched_by"] == "bench.py" and d["config"]["objects_per_gpu"] == 48
    assert d["config"]["collective_by"] == "torch.distributed"
    assert d["parity"]["pass"] and d["parity_checked_objects"] == 8
    g = d["gather_cost"]
    assert g["bytes_sent_per_rank"] == 48 * 86 * 513 * 4 and g["bytes_received_per_rank"] == g["bytes_sent_per_rank"]
    assert g["ms_per_step_without_gather"] > 0 and g["value_without_gather"] > 0
    mx = d["mix"]
    assert mx["bytes_per_rank"] == 86 * 513 * 4 and mx["value"] > 0 and mx["objects_total"] == 96
    gr = d["gather_to_root"]
    assert gr["bytes_received_by_root"] == 48 * 86 * 513 * 4 and gr["value"] > 0 and gr["objects_total"] == 96
    # every rank got its share of the host's cores for planning, and says whether the host was the bottleneck
    assert 1 <= d["config"]["host_planner_threads"] <= max(1, d["config"]["host_cores"] // 2 - 1)
    assert isinstance(d["timing"]["host_bound"], bool)
    s = d["weak"]
    assert s["scaling"] == "weak" and s["objects_total"] == 192 and s["objects_rank0"] == 96 and s["gather"] is True
    # whole-job value: the job's objects over the slowest rank's time
    assert abs(d["value"] - 96 * 86 * 513 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]


@pytest.mark.gpu
def test_one_rank_under_torchrun_issues_the_groups_collectives_on_a_one_rank_communicator():
    """the launcher the driver uses for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), with N = 1 and
    PBSO_BENCH_GATHER_SELF=1: backend nccl (= RCCL), and the device group in its PBSO_GROUP_RCCL_ALWAYS transport -- librccl
    loaded, ncclCommInitRank for ONE rank, and every leg's collective ISSUED on it (in-place ncclAllGather; ncclSend / ncclRecv to
    itself; ncclAllReduce) beside the next step's oscillator bank.  What it cannot show is traffic: with one rank nothing crosses a
    link (bytes_received_per_rank == 0); the ranks' LOGIC at world 2, 3, 8 is tests/test_group.py's loopback transport"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PBSO_BENCH_GATHER_SELF="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PBSO_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--objects", "256", "--steps", "8",
                        "--warmup", "2", "--settle", "4", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["gather"] is True and d["config"]["backend"] == "nccl" and d["config"]["rccl_ranks"] == 1
    assert d["config"]["launched_by"] == "torch.distributed.run" and d["parity"]["pass"]
    assert d["config"]["collective_by"].startswith("pbso_group")          # the collective was issued by the C++ device group
    assert d["config"]["buffers_per_step"] == 860                          # (the driver's command: the default step, ten seconds of audio)
    assert d["gather_cost"]["bytes_sent_per_rank"] == 256 * 860 * 513 * 4 and d["gather_cost"]["bytes_received_per_rank"] == 0
    assert "RCCL all-gather" in d["config"]["workload"] and d["gather_to_root"]["value"] > 0 and d["mix"]["value"] > 0
