"""bench.py's multi-rank plumbing on CPU: `python bench.py --gpus 2` (as the driver calls it, no torchrun environment)
re-launches itself as two ranks; tests/bench_cpu_launch.py calls bench.main() with a stand-in backend (gloo + the fiber emulator
and a toy network), so the launcher, the r::world sharding, weak/strong batch arithmetic, the single all-reduce and the JSON
contract are exercised without a GPU.  Numbers are meaningless here; the fields are not."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "tests", "bench_cpu_launch.py"), "--workload", "toy", "--steps", "2", "--warmup", "1"] + list(flags),
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                       # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_launches_two_ranks_by_itself(scaling):
    out = _run("--gpus", "2", "--scaling", scaling)
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["unit"] == "frames/s" and out["value"] > 0
    c = out["config"]
    assert c["collective_world_size"] == 2 and c["collective_backend"] == "gloo" and c["allreduce_ms"] > 0
    if scaling == "weak":
        assert c["per_gpu_batch"] == 4 and c["global_batch"] == 8
    else:
        assert c["global_batch"] == 8 and c["per_gpu_batch"] == 4
    assert c["frames_per_step"] == c["global_batch"] * 13            # T = 13 real frames per utterance, all ranks counted
    assert out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True


def test_single_rank_line_has_no_collective_fields():
    out = _run()
    assert out["n_gpus"] == 1 and "collective_world_size" not in out["config"] and out["config"]["global_batch"] == 4
