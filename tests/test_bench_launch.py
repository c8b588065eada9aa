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


def _run(*flags, workload="toy"):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "tests", "bench_cpu_launch.py"), "--workload", workload, "--steps", "2", "--warmup", "1"] + list(flags),
                       env=env, capture_output=True, text=True, timeout=900)
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
    assert out["self_check"]["cluster_reserve"] == 0 and out["self_check"]["rank_r_holds_utterances_r_mod_world"] is True      # (no overlapped exchange: nothing reserved)


def test_single_rank_line_has_no_collective_fields():
    out = _run()
    assert out["n_gpus"] == 1 and "collective_world_size" not in out["config"] and out["config"]["global_batch"] == 4


def test_world_8_launch_is_boring():
    """The first 8-rank run (round-5 verdict, ask 6): `bench.py --gpus 8` as the driver calls it, eight gloo ranks on the emulator —
    the collective spans exactly eight ranks, every rank was fed utterances r::8 of the global minibatch (fingerprints gathered and
    compared on rank 0), the weak line counts all ranks' frames, and the `strong` sub-object shards global batch 128 into 16 per
    rank and carries BOTH one-GPU forms it is quoted against."""
    out = _run("--gpus", "8", workload="toy_strong128")
    c, chk = out["config"], out["self_check"]
    assert out["n_gpus"] == 8 and out["scaling"] == "weak"
    assert c["collective_world_size"] == 8 and c["parallelism"] == "dp8" and c["per_gpu_batch"] == 4 and c["global_batch"] == 32
    assert c["frames_per_step"] == 32 * 13
    assert chk["ok"] is True and chk["collective_world_size_is_n_gpus"] is True and chk["rank_r_holds_utterances_r_mod_world"] is True
    assert chk["steps_skipped"] == 0 and chk["cluster_aborts"] == 0
    st = out["strong"]
    assert st["global_batch"] == 128 and st["per_gpu_batch"] == 16 and st["scaling"] == "strong" and st["value"] > 0
    for k in ("one_gpu_ms_per_step_encoder_in_passes", "one_gpu_ms_per_step_encoder_step_kernels", "speedup_vs_one_gpu",
              "speedup_vs_one_gpu_encoder_step_kernels"):
        assert st[k] > 0, k
    assert st["one_gpu_ms_per_step"] == min(st["one_gpu_ms_per_step_encoder_in_passes"], st["one_gpu_ms_per_step_encoder_step_kernels"])


def test_overlapped_allreduce_reserves_cus_for_the_collective():
    """`--overlap-allreduce` with two ranks: two buckets, and the cluster launches leave Trainer.OVERLAP_RESERVE CUs to the
    collective's work-groups (the knob took effect in the library the step ran on)."""
    out = _run("--gpus", "2", "--overlap-allreduce")
    assert out["config"]["overlap_allreduce"] is True and out["self_check"]["cluster_reserve"] == 32
    assert out["self_check"]["ok"] is True and out["self_check"]["rank_r_holds_utterances_r_mod_world"] is True
