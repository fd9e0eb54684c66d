"""bench.py's launch layer: `--gpus N` must measure N ranks however the script is started.
CPU: the self-launch, the rendezvous and the collectives of an N-rank run with the device work left out (gloo);
GPU (one device): two ranks sharing device 0 (gloo: RCCL refuses two ranks on one device), the strong-scaling cut and the
single-process mode over nyx_hip_propagate_batch_sharded."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "NYX_BENCH_SELF_LAUNCHED"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("n", [2, 3])
def test_self_launch_starts_n_ranks(n):
    p, line = _run(["--gpus", str(n), "--selftest-launch"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == n and line["ranks"] == n and line["self_launched"] is True
    assert line["gathered_sum"] == 28.0 * sum(range(1, n + 1))        # every rank's piece arrived at rank 0
    assert line["max_time"] == 0.5 + (n - 1)                           # max over ranks
    assert line["per_rank_kernel_ms"] == [10.0 * (r + 1) for r in range(n)]
    assert len([ln for ln in p.stdout.splitlines() if ln.startswith("{")]) == 1   # ONE line, from rank 0


def test_a_failing_rank_ends_the_run_instead_of_hanging_it():
    """The launcher polls ALL its ranks: rank 1 dies before the rendezvous, rank 0 waits in it - the run must end with rank 1's code."""
    import time
    t0 = time.time()
    p, line = _run(["--gpus", "2", "--selftest-launch"], env={"NYX_BENCH_SELFTEST_FAIL_RANK": "1"}, timeout=120)
    assert p.returncode == 3 and line is None and time.time() - t0 < 60


def test_world_size_must_agree_with_gpus():
    p, line = _run(["--gpus", "2", "--selftest-launch"], env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and line is None and "WORLD_SIZE=3" in p.stderr


def test_stale_traffic_is_not_quoted(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    from kernel_stamp import kernel_source_stamp
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    import kernel_stamp
    real = kernel_source_stamp()
    monkeypatch.setattr(kernel_stamp, "kernel_source_stamp", lambda root=None: real)
    rec = {"config": 2, "n": 10000, "hours": 24.0, "degree": 70, "hbm_bytes_per_launch": 1.0e11}
    (prof / "roundXX_cfg2_hbm_traffic.json").write_text(json.dumps(rec))                       # unstamped: a pass of an unknown build
    t, src = bench.measured_traffic(2, 10000, 24.0, 70)
    assert t is None and src.startswith("stale:")
    (prof / "roundXY_cfg2_hbm_traffic.json").write_text(json.dumps(dict(rec, kernel_source_stamp="0123456789abcdef")))
    t, src = bench.measured_traffic(2, 10000, 24.0, 70)
    assert t is None and "0123456789abcdef" in src
    (prof / "roundXZ_cfg2_hbm_traffic.json").write_text(json.dumps(dict(rec, kernel_source_stamp=real)))
    t, src = bench.measured_traffic(2, 10000, 24.0, 70)
    assert t == 1.0e11 and src.endswith("roundXZ_cfg2_hbm_traffic.json")
    assert bench.measured_traffic(3, 5000, 720.0, 0) == (None, None)


SMALL = ["--n", "256", "--hours", "0.25", "--degree", "8", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-dense-output", "--no-host-call"]


@pytest.mark.gpu
def test_two_ranks_on_one_device():
    p, line = _run(["--gpus", "2", "--oversubscribe", "--backend", "gloo", *SMALL])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["launch"] == "self-launched ranks" and line["backend"] == "gloo"
    assert line["config"]["trajectories_total"] == 512 and len(line["per_rank_kernel_ms"]) == 2 and line["all_gather_ms"] is not None
    assert line["value"] > 0 and line["scaling"] == "weak"


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [[], ["--config", "4", "--n", "64"]])
def test_the_rccl_path_runs_with_one_rank(cfg):
    """The N > 1 launches of the driver are the first time RCCL sees this script; `--force-collectives` issues every collective of
    a step (all-gather of the final states, all-reduce of the 55 moments, the reductions of the report) on a one-rank nccl group,
    on the tensors and streams the N-rank run uses.  Plain kernel and the covariance-mapping loop."""
    p, line = _run(["--gpus", "1", "--force-collectives", *SMALL, *cfg])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 1 and line["launch"] == "single rank, collectives forced"
    assert line["backend"] == "nccl" and line["rccl_ranks"] == 1 and line["all_gather_ms"] is not None and line["all_gather_ms"] >= 0.0
    if not cfg:
        assert line["ensemble_moments"]["count"] == 256.0 and line["ensemble_moments"]["trace_cov_pos_km2"] > 0.0


@pytest.mark.gpu
def test_strong_scaling_cuts_one_ensemble():
    p, line = _run(["--gpus", "2", "--oversubscribe", "--backend", "gloo", "--scaling", "strong", *SMALL[:1], "257", *SMALL[2:]])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["scaling"] == "strong" and line["config"]["trajectories_total"] == 257 and line["config"]["trajectories_per_gpu"] == 129


@pytest.mark.gpu
def test_single_process_mode_drives_the_sharded_entry():
    p, line = _run(["--gpus", "2", "--oversubscribe", "--single-process", *SMALL])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["launch"] == "single-process" and line["contexts"] == 2 and line["config"]["trajectories_total"] == 512
    assert len(line["per_rank_kernel_ms"]) == 2 and line["value"] > 0


@pytest.mark.gpu
def test_rccl_refuses_shared_device_with_a_message():
    import torch
    if torch.cuda.device_count() > 1:
        pytest.skip("needs a box where two ranks must share a device")
    p, line = _run(["--gpus", "2", "--oversubscribe", *SMALL])
    assert p.returncode != 0 and line is None
