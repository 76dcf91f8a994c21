"""GPU: the N > 1 path of bench.py rehearsed on a one-GPU box.  `python bench.py --gpus 2` -- exactly how the driver starts the scaling
runs -- launches its own two ranks; with SPGAN_DIST_BACKEND=gloo both share device 0 and the flat gradient all-reduces go through gloo's
CUDA path, so the whole data-parallel schedule (four captured graphs per rank, the generator's forward replayed under D's all-reduce,
rank-synchronous Adam, barrier + max-over-ranks timing, ONE JSON line from rank 0) runs on the HIP kernels.  A functional check: the
line says "rehearsal", its value is not a measurement (two processes time-slice one GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_rank_schedule_on_one_gpu():
    env = dict(os.environ, SPGAN_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SPGAN_BENCH_SELFTEST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["world_size_observed"] == 2 and line["collective_backend"] == "gloo"
    assert line["config"]["global_batch"] == 64 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    assert line["hipgraph_replay"] is True, "the captured data-parallel schedule must not have fallen back to eager issue"
    assert "rehearsal" in line and line["value"] > 0 and line["steps"] == 2
    assert "falling back to eager" not in r.stderr
    # first contact (round-4 review item 7): both ranks seen by the first collective, the overlapped default bit-equal to the sequential
    # schedule on the HIP kernels and on both ranks, the sequential schedule measured first, no fallback taken
    fc = line["first_contact"]
    assert line["rccl_ranks_seen"] == 2 and fc["schedules_bit_equal"] is True and fc["ranks_bit_equal"] is True
    assert line["dp_schedule"] == "overlapped" and "dp_fallback" not in line and line["dp_sequential_schedule"]["ms_per_step"] > 0
