"""GPU: the C++ host mirror of the reference's Go API (dieselfluid_amd/host) runs the
reference's own tests (TestGPUCompile, TestOpenCompute) and the drivers' channel protocol
through libdslsph.so; values are checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "dieselfluid_amd", "host", "demo")


def _run():
    if not os.path.exists(DEMO):
        subprocess.check_call(["make", "-C", os.path.dirname(DEMO), "-s"])
    out = subprocess.run([DEMO], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    res = {}
    for line in out.stdout.splitlines():
        tok = line.split()
        if len(tok) >= 3 and tok[0] not in ("demo",):
            res[tok[0]] = dict(zip(tok[1::2], tok[2::2]))
    assert "demo ok" in out.stdout
    return res


def test_host_mirror_runs_reference_tests():
    r = _run()
    # TestGPUCompile: zero-length origin -> every particle at the origin, Init returns
    assert int(r["TestGPUCompile"]["n"]) == 4096 and float(r["TestGPUCompile"]["max_abs_pos"]) == 0.0
    # sph.Init(.., 16, true): delta (host scalar) and the Init-time density pass
    prm = po.params_reference(16)
    prm.neigh_mode = po.NEIGH_GRID
    ora = po.OracleSPH.init(prm, pci=True)
    assert np.float32(r["Init16"]["delta"]) == np.float32(ora.delta)
    assert abs(float(r["Init16"]["rho_mean"]) - float(ora.densities().mean())) < 1e-5 * float(ora.densities().mean())
    assert float(r["Init16"]["rho0"]) == 512.0
    assert abs(float(r["Init16"]["f1y"]) - float(ora.forces()[0, 1])) < 1e-4
    # TestOpenCompute: one PCISPH step completes, then QUIT is honoured
    assert r["TestOpenCompute"]["first_message"] == "SAMPLER_UPDATE" and int(r["TestOpenCompute"]["steps"]) >= 1
    # WCSPH.Run_ with the THREAD_GO handshake: free-fall known answer after 3 steps
    assert int(r["WCSPHFreeFall"]["steps"]) == 3
    assert np.float32(r["WCSPHFreeFall"]["vy"]) == np.float32(-0.5886)
    assert np.float32(r["WCSPHFreeFall"]["y0"]) == np.float32(-1.0117719)
    g = r["GPUPredictorCorrector"]
    assert g["err"] == "''" and int(g["refresh"]) == 2 and int(g["npos"]) == 512 * 3
    assert int(g["valid"]) == 1 and int(g["bad_buffer_error"]) == 1
