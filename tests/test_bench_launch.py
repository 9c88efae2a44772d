"""`python bench.py --gpus N` with no launcher around it (the form the driver's scaling run may use) must start its own ranks:
the script re-executes itself under torch.distributed.run.  Checked here on CPU with --launch-check (gloo, no GPU work): both ranks
start, join one group and agree on the contiguous block ranges of dist.shard_range; rank 0 alone prints the line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_2_launches_itself_and_ranks_agree_on_the_shards():
    r = _run(["--gpus", "2", "--launch-check", "--blocks", "2049"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["launch_check"] and line["world"] == 2
    assert line["ranges"] == [[0, 0, 1025], [1, 1025, 2049]]           # uneven: the first rank takes the odd block
    assert "torch.distributed.run" in r.stderr                         # it did go through the launcher


def test_world_size_mismatch_is_an_error_message_not_an_assert():
    r = _run(["--gpus", "2", "--launch-check"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr and "AssertionError" not in r.stderr
