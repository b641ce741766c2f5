"""`python bench.py --gpus N` starts its own ranks (one process per GPU, contiguous shards, one RCCL
all-reduce of the objective / gradient): the command line it would run is checked on the CPU; on the GPU box the 1-rank
rehearsal goes through the same launcher and must agree with the plain single-process run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(argv, extra_env=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CMAD_BENCH_FORCE_DIST")}
    env.update(extra_env or {})
    res = subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, env=env, timeout=timeout)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 8])
def test_dry_launch_prints_one_rank_per_gpu(n):
    out = run_bench(["--gpus", str(n), "--steps", "7", "--warmup", "2", "--dry-launch"])
    cmd = out["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert f"--nproc-per-node={n}" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", str(n), "--steps", "7", "--warmup", "2"]       # the ranks get the caller's arguments, not --dry-launch


def test_a_launched_rank_does_not_launch_again():
    """Under a launcher (WORLD_SIZE set) bench.py must go straight to the measurement: with no GPU here that is the
    `needs a GPU` assertion, not another torch.distributed.run."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, env=env, timeout=600)
    import torch
    if not torch.cuda.is_available():
        assert res.returncode != 0 and "needs a GPU" in res.stderr
        assert "launch" not in res.stdout


@pytest.mark.gpu
def test_one_rank_rehearsal_through_the_launcher_matches_the_plain_run():
    args = ["--gpus", "1", "--steps", "10", "--warmup", "3", "--points", "10000000", "--no-cpu-baseline"]
    plain = run_bench(args)
    dist = run_bench(args, {"CMAD_BENCH_FORCE_DIST": "1"})
    assert plain["n_gpus"] == 1 and plain["rccl"]["world_size"] == 1 and plain["rccl"]["collective_calls"] == 0
    assert dist["n_gpus"] == 1
    assert dist["rccl"]["world_size"] == 1 and dist["rccl"]["backend"] == "nccl"
    assert dist["rccl"]["launcher"] == "torch.distributed.run"
    assert dist["rccl"]["collective_calls"] == 10 and dist["rccl"]["payload_doubles"] == 12
    # configs[4]'s workload rides along: (J, grad) = 13 doubles through the collective, once per evaluation
    assert dist["objective"]["rccl"] == {"collective_calls": 10, "payload_doubles": 13}
    assert dist["objective"]["total_points"] == 10000000
    # same kernels on the same box, plus one all-reduce per ~0.4 ms step whose latency (tens of microseconds through RCCL) is not
    # hidden on ten steps: the rehearsal must be in the same range, not equal (this is a bookkeeping test, not a measurement)
    for a, b in ((plain["value"], dist["value"]), (plain["objective"]["value"], dist["objective"]["value"])):
        assert 0.5 * a < b < 1.5 * a, (a, b)


@pytest.mark.gpu
def test_two_rank_rehearsal_sharing_one_gpu():
    """The N > 1 path of `bench.py` end to end on a one-GPU box: `--gpus 2` starts two ranks through its own launcher, both
    compute on device 0 (CMAD_BENCH_SHARED_GPU=1; collectives over gloo, since RCCL refuses two ranks on one device), shard the
    batch, meet at the barriers, reduce the timing over the ranks, and rank 0 alone prints the line.  Checks the bookkeeping --
    world size, collectives counted, per-rank kernel times, both records -- not the numbers (the ranks share the card)."""
    out = run_bench(["--gpus", "2", "--steps", "10", "--warmup", "3", "--points", "2000000", "--no-cpu-baseline"],
                    {"CMAD_BENCH_SHARED_GPU": "1"})
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["rccl"]["world_size"] == 2 and out["rccl"]["backend"] == "gloo" and "rehearsal" in out["rccl"]
    assert out["rccl"]["launcher"] == "torch.distributed.run"
    assert out["rccl"]["collective_calls"] == 10 and out["rccl"]["payload_doubles"] == 12
    assert len(out["timeline"]["per_rank_kernel_ms"]) == 2 and all(t > 0 for t in out["timeline"]["per_rank_kernel_ms"])
    assert out["config"]["points_per_gpu"] == 2000000
    # whole-job aggregate: both ranks' points over the max-over-ranks time
    assert abs(out["value"] - 2 * 2000000 / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    assert out["objective"]["total_points"] == 4000000 and out["objective"]["rccl"] == {"collective_calls": 10, "payload_doubles": 13}
    assert "cpu_baseline" not in out or out["cpu_baseline"] is None          # rank 0 at N = 1 only
