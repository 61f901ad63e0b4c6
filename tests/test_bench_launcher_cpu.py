"""`python bench.py --gpus N` as the driver invokes it (no torchrun, no WORLD_SIZE): the parent must start the N ranks
itself without touching the GPU, forward rank 0's ONE JSON line and the children's exit code.
Rehearsed on CPU with --dry-run: gloo ranks shard a stub grid through misti_amd.dist.evaluate_sharded and gather it."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_starts_its_own_ranks(world):
    r = _run(["--gpus", str(world), "--dry-run", "--steps", "3", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line on stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["config"]["world_size"] == world
    assert j["data"] == "dry-run" and j["gather_equals_unsharded"] is True
    assert j["steps"] == 3


def test_single_rank_dry_run_needs_no_launcher():
    r = _run(["--dry-run", "--steps", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip())
    assert j["n_gpus"] == 1 and j["gather_equals_unsharded"] is True


def test_a_failing_rank_fails_the_launcher():
    """Without a GPU the real (non dry-run) ranks exit non-zero: the parent must report that, not a JSON line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_parent_does_not_import_torch_before_launching():
    """The launching parent must never initialise the GPU: bench.py may import torch only inside main(), after launch_ranks."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main():")]
    assert "import torch" not in head.replace("import torch.distributed as dist", "").replace("    import torch", "")
    body = src[src.index("def main():"):]
    assert body.index("launch_ranks(a)") < body.index("import torch")
