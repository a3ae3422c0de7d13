"""bench.py's output contract (one JSON line per run; the driver parses it) and its refusal to run without a GPU."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=600):
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                          cwd=ROOT, env=env)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU refusal")
def test_bench_refuses_to_run_without_a_gpu():
    r = _run("--steps", "1", "--warmup", "0", timeout=300)
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stderr + r.stdout)
    assert not any(line.startswith("{") for line in r.stdout.splitlines())       # no bench line from a machine without the GPU


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the refusal on a machine with fewer GPUs than asked for")
def test_bench_gpus_n_fails_loudly_instead_of_running_fewer_ranks():
    """`--gpus 2` must start two ranks or fail: never a line that says n_gpus = 2 (or 1) from a single process.  Without a
    launcher bench.py spawns the ranks itself and refuses when the devices are not there; under a launcher whose WORLD_SIZE
    disagrees with --gpus it refuses as well."""
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "refusing" in r.stderr
    assert not any(line.startswith("{") for line in r.stdout.splitlines())
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert not any(line.startswith("{") for line in r.stdout.splitlines())


@pytest.mark.gpu
def test_bench_spawn_path_runs_the_ranks_it_promises():
    """`--gpus 1 --spawn`: the torch.distributed.run child every N > 1 run goes through, on the one GPU of this box -- RCCL process
    group, ranks counted by an all-reduce, one line with n_gpus = ranks_joined = 1."""
    r = _run("--gpus", "1", "--spawn", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-secondary")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["ranks_joined"] == 1 and d["value"] > 1e5


@pytest.mark.gpu
def test_bench_line_contract():
    r = _run("--steps", "5", "--warmup", "1", "--cpu-repeats", "1", "--no-secondary")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict),
                 ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["steps"] == 5 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "decode-steps" in d["metric"] and d["unit"] == "decode-steps/s" and "workload" in d["config"]
    assert d["value"] == pytest.approx(64 * 20 / (d["ms_per_step"] * 1e-3), rel=1e-3)          # B x T per timed step
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and "traffic" in rf
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=2e-3) and 0.05 < rf["frac"] < 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and isinstance(cb["sample"], str)
    assert d["value"] > 20 * cb["value"]                                                        # sanity: the GPU path is the one measured
