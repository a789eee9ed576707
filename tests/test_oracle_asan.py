"""The oracle's main paths (sph.Init in lsh_ref mode, both reference loops, the grid mode with the
build-defined terms, the field operators) under AddressSanitizer + UndefinedBehaviorSanitizer.
CPU only (sanitizers run on the CPU build; GPU ASan is not available)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_runs_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan-run"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "asan driver ok" in r.stdout
