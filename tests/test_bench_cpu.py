"""bench.py's plumbing on the CPU (SPGAN_BENCH_SELFTEST=1: tiny shapes, gloo, kernel-model doubles): `python bench.py --gpus 2`
started WITHOUT torchrun -- exactly how the driver starts it -- must launch its own two ranks, run the data-parallel step,
print ONE JSON line from rank 0 with n_gpus = the world size it observed, and exit 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = dict(os.environ, SPGAN_BENCH_SELFTEST="1", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=600)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("n", [1, 2])
def test_bench_self_launches_its_ranks(n):
    r = _run(["--gpus", str(n), "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["selftest"] is True and line["n_gpus"] == n and line["world_size_observed"] == n
    assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak" and line["unit"] == "shapes/s"
    assert line["config"]["global_batch"] == 2 * n and line["config"]["parallelism"] == "dp%d" % n
    assert line["value"] > 0 and line["ms_per_step"] > 0
    if n > 1:
        assert line["collective_backend"] == "gloo"


def test_bench_eight_ranks():
    """`python bench.py --gpus 8` -- the driver's scaling run -- as a plumbing self-test: eight gloo ranks on this host, the line
    reports world size 8, global batch 8 x the per-rank batch, weak scaling."""
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "1"], {"OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["selftest"] is True and line["n_gpus"] == 8 and line["world_size_observed"] == 8
    assert line["config"]["global_batch"] == 16 and line["config"]["parallelism"] == "dp8" and line["scaling"] == "weak"


def test_bench_refuses_a_mismatched_world():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.parametrize("cfg", ["c4", "c5"])
def test_bench_config_switch(cfg):
    """`--config c4 / c5` (the per-GPU shapes of BASELINE configs[3] / configs[4]) are accepted, named in the line, and the default (the
    driver's command) stays c2; an unknown config is refused before anything runs."""
    r = _run(["--config", cfg, "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_lines(r.stdout)[0]
    assert line["bench_config"] == cfg and line["selftest"] is True
    assert line["gf_per_shape_step_reference"] == (67.3 if cfg == "c4" else 32.6)
    assert _run(["--config", "c9"]).returncode != 0


def test_bench_first_contact_report_and_fallbacks():
    """Round-4 review item 7: on N > 1 ranks bench.py (1) all-reduces one float per rank under a watchdog and reports the ranks seen,
    (2) runs one sequential and one overlapped eager step from identical state and requires bit-equal parameters across schedules and
    ranks, (3) measures the sequential schedule in full BEFORE the overlapped default, so that a hang of the latter still leaves a
    measured line with a "dp_fallback" key (here: a simulated hang and a simulated wrong result of the overlapped schedule)."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_lines(r.stdout)[0]
    fc = line["first_contact"]
    assert line["rccl_ranks_seen"] == 2 and fc["schedules_bit_equal"] is True and fc["ranks_bit_equal"] is True
    assert line["dp_schedule"] == "overlapped" and "dp_fallback" not in line and line["dp_sequential_schedule"]["shapes_per_s"] > 0
    # the overlapped schedule hangs: the watchdog prints the sequential measurement and every rank leaves with status 0
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"SPGAN_BENCH_TEST_HANG": "overlap", "SPGAN_BENCH_WATCHDOG_S": "20"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["dp_schedule"] == "sequential" and "did not complete" in lines[0]["dp_fallback"] and lines[0]["value"] > 0
    assert lines[0]["n_gpus"] == 2 and lines[0]["first_contact"]["rccl_ranks_seen"] == 2
    # the overlapped schedule computes something else: the sequential one is what gets timed, and the line says why
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"SPGAN_BENCH_TEST_BREAK": "overlap"})
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_lines(r.stdout)[0]
    assert line["dp_schedule"] == "sequential" and "not bit-equal" in line["dp_fallback"] and line["first_contact"]["schedules_bit_equal"] is False
    # SPGAN_DP_OVERLAP=0: the sequential schedule by request
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"SPGAN_DP_OVERLAP": "0"})
    line = _json_lines(r.stdout)[0]
    assert r.returncode == 0 and line["dp_schedule"] == "sequential" and "SPGAN_DP_OVERLAP=0" in line["dp_fallback"]
