"""bench.py's launch contract: `--gpus N` means N ranks -- started by bench.py itself when no
launcher is around it -- and a world size that is not the one asked for is an error, never a silent 1-rank run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(kw)
    return e


def test_world_size_mismatch_exits_2_before_touching_the_gpu():
    # a launcher that started 3 ranks for a run that asks for 2: refuse (checked before torch is even imported)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2, (r.returncode, r.stderr[-400:])
    assert "WORLD_SIZE=3" in r.stderr
    # ... and the default --gpus 1 under a 2-rank launcher likewise
    r = subprocess.run([sys.executable, BENCH], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2, (r.returncode, r.stderr[-400:])


@pytest.mark.gpu
def test_gpus_2_without_a_launcher_runs_two_ranks():
    """`python bench.py --gpus 2` (no torchrun around it): bench.py starts the two ranks as a child launcher; rehearsed
    on one card over gloo (DSL_BENCH_BACKEND), the library's slab driver on a host-staged transport."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--n3", "96", "--steps", "3", "--warmup", "2"],
                       env=_env(DSL_BENCH_BACKEND="gloo", DSL_BENCH_WATCHDOG_S="240"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen_by_rccl"] == 2
    assert out["slab_overflow"] == 0 and out["value"] > 0
