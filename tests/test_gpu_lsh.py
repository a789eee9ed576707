"""GPU, DSL_NEIGH_LSH_REF: the reference's own neighbour rule (sampler/lsh/lsh.go -- 255
buckets, 100 samples with duplicates and non-neighbours) on the device.  Same sample order
and the same float32 arithmetic as the oracle's lsh_ref mode, so densities must agree bit for
bit; terms that go through pow() may differ in the last place."""
import os

import numpy as np
import pytest

import helpers
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _lsh_engine(n3, hv):
    from dieselfluid_amd import SPHEngine
    from dieselfluid_amd.engine import reference_params
    p = reference_params(n3)
    p.neigh_mode = 0  # DSL_NEIGH_LSH_REF
    p.math_mode = 0
    eng = SPHEngine(p)
    eng.set_hash_vectors(hv)
    return eng, p


@pytest.mark.parametrize("n3", [8, 16])
def test_init_sequence_matches_fixture_bit_for_bit(n3):
    """sph.Init's passes (fluid.go:72-75) on the lattice with the fixture's hash vectors."""
    from dieselfluid_amd import scenes
    z = np.load(os.path.join(G, "init_lsh_ref.npz"))
    eng, p = _lsh_engine(n3, z["hash_vectors"])
    eng.upload("positions", scenes.lattice_positions(n3))
    eng.nn()                                                   # UpdateSampler
    eng.density_all()                                          # DensityAll
    eng.external_all(np.array([0, -9.81, 0], np.float32))      # ExternalAll
    eng.viscous_all()                                          # ViscousAll
    assert np.array_equal(_bits(eng.download("densities")), _bits(z[f"n{n3}_densities"]))
    f, want = eng.download("forces"), z[f"n{n3}_forces"]
    assert np.array_equal(np.isnan(f), np.isnan(want))
    assert np.array_equal(_bits(np.nan_to_num(f)), _bits(np.nan_to_num(want)))


def test_flattened_table_matches_getdata1d():
    """HashSampler.GetData1D (lsh.go:70-80) and the per-bucket sample lists."""
    hv = po.default_hash_vectors(11)
    n3 = 12
    pos = helpers.jittered_lattice(n3, 0.3)
    eng, p = _lsh_engine(n3, hv)
    eng.upload("positions", pos)
    eng.nn()
    prm = po.params_reference(n3)
    ora = po.OracleSPH.from_state(prm, pos, hash_vectors=hv)
    assert np.array_equal(eng.lsh_table(), ora.lsh_data_1d())


def test_wcsph_and_passes_match_oracle_lsh_mode():
    hv = po.default_hash_vectors(5)
    n3 = 12
    pos = helpers.jittered_lattice(n3, 0.25)
    vel = helpers.seeded_velocities(n3 ** 3, 0.1)
    eng, p = _lsh_engine(n3, hv)
    eng.upload("positions", pos)
    eng.upload("velocities", vel)
    prm = po.params_reference(n3)
    ora = po.OracleSPH.from_state(prm, pos, vel=vel, hash_vectors=hv)
    eng.density_all(); ora.density_all()
    assert np.array_equal(_bits(eng.download("densities")), _bits(ora.densities()))
    eng.viscous_all(); ora.viscous_all()
    fa, fb = eng.download("forces"), ora.forces()
    assert np.array_equal(np.isnan(fa), np.isnan(fb))
    assert np.array_equal(_bits(np.nan_to_num(fa)), _bits(np.nan_to_num(fb)))
    eng.gradient_pressure_force(); ora.gradient_pressure_force()
    fa, fb = np.nan_to_num(eng.download("forces")), np.nan_to_num(ora.forces())
    assert helpers.rel_err(fa, fb) < 1e-6                       # f64 pow: device libm vs glibc
    # the reference WCSPH loop (no pressure force): exact
    eng2, _ = _lsh_engine(n3, hv)
    eng2.upload("positions", pos)
    eng2.upload("velocities", vel)
    ora2 = po.OracleSPH.from_state(prm, pos, vel=vel, hash_vectors=hv)
    eng2.wcsph_step(3); ora2.wcsph_step(3)
    assert np.array_equal(_bits(eng2.download("positions")), _bits(ora2.positions()))
    assert np.array_equal(_bits(eng2.download("densities")), _bits(ora2.densities()))


def test_pcisph_step_matches_oracle_lsh_mode():
    hv = po.default_hash_vectors(9)
    n3 = 12
    pos = helpers.jittered_lattice(n3, 0.1)
    vel = helpers.seeded_velocities(n3 ** 3, 0.05)
    eng, p = _lsh_engine(n3, hv)
    p.delta = 1.0e-4
    eng.set_params(p)
    eng.upload("positions", pos)
    eng.upload("velocities", vel)
    prm = po.params_reference(n3)
    ora = po.OracleSPH.from_state(prm, pos, vel=vel, hash_vectors=hv)
    ora.delta = 1.0e-4
    eng.pcisph_begin(); ora.pcisph_begin()
    for _ in range(2):
        eng.pcisph_step(1); ora.pcisph_step(1)
        assert eng.stats().pci_iters == ora.pci_iters
        a, b = eng.download("positions"), ora.positions()
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert helpers.rel_err(np.nan_to_num(a), np.nan_to_num(b)) < 1e-5


def test_lsh_mode_rejects_fast_math_and_slabs():
    from dieselfluid_amd import DslError, SPHEngine
    from dieselfluid_amd.engine import reference_params
    p = reference_params(4)
    p.neigh_mode = 0
    p.math_mode = 1
    with pytest.raises(DslError):
        SPHEngine(p)
    p.math_mode = 0
    eng = SPHEngine(p)
    with pytest.raises(DslError):
        eng.slab_config(2, 0.0, 1.0)
