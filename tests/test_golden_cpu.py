"""CPU: the oracle reproduces the committed golden fixtures bit for bit (guards the
checker itself against compiler/flag drift), and the product-side host generators agree
with the oracle's independent C restatements."""
import os

import numpy as np

import helpers
from oracle import pyoracle as po

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_init_lsh_ref_fixture():
    z = np.load(os.path.join(G, "init_lsh_ref.npz"))
    assert np.array_equal(z["hash_vectors"], po.default_hash_vectors(7))
    for n3 in (8, 16):
        s = po.OracleSPH.init(po.params_reference(n3), hash_vectors=z["hash_vectors"], pci=True)
        assert np.array_equal(_bits(s.densities()), _bits(z[f"n{n3}_densities"]))
        assert np.array_equal(_bits(s.forces()), _bits(z[f"n{n3}_forces"]))
        assert np.float32(s.delta) == z[f"n{n3}_delta"]
        assert np.array_equal(s.get_samples(0), z[f"n{n3}_samples0"])


def test_reference_grid_fixture():
    from dieselfluid_amd import scenes
    z = np.load(os.path.join(G, "reference_grid_n12.npz"))
    p, _ = scenes.reference_scene(12)
    q = helpers.oracle_params(p)
    assert np.array_equal(z["positions"], helpers.jittered_lattice(12, 0.2))
    s = po.OracleSPH.from_state(q, z["positions"], vel=z["velocities"])
    s.density_all()
    assert np.array_equal(_bits(s.densities()), _bits(z["densities"]))
    s.viscous_all()
    assert np.array_equal(_bits(s.forces()), _bits(z["viscous_force"]))
    for iters in (5, 4):
        q.pci_max_iters = iters
        s3 = po.OracleSPH.from_state(q, z["pci_positions0"], vel=z["pci_velocities0"])
        s3.delta = 1.0e-4
        s3.pcisph_begin()
        s3.pcisph_step(1)
        assert np.array_equal(_bits(s3.positions()), _bits(z[f"pci{iters}_positions"]))
        assert s3.pci_iters == int(z[f"pci{iters}_iters"])


def test_dambreak_fixture():
    from dieselfluid_amd import scenes
    z = np.load(os.path.join(G, "dambreak_n12.npz"))
    p, pos = scenes.dambreak_scene(12)
    assert np.array_equal(pos, z["positions0"])
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (12 ** 3, 1))
    s = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
    s.wcsph_step(1)
    assert np.array_equal(_bits(s.positions()), _bits(z["x1"]))
    s.wcsph_step(9)
    assert np.array_equal(_bits(s.positions()), _bits(z["x10"]))
    assert np.array_equal(_bits(s.velocities()), _bits(z["v10"]))
    assert not np.isnan(z["x10"]).any()


def test_host_scene_generators_match_oracle():
    """Row I: the product's numpy lattice / dam-break generators vs the oracle's C ones."""
    from dieselfluid_amd import scenes
    for n3 in (4, 16, 20):
        assert np.array_equal(scenes.lattice_positions(n3), po.lattice_positions(n3))
    assert np.all(scenes.lattice_positions(8, origin=()) == 0)
    for n3, dx in ((8, 0.125), (20, 0.05)):
        assert np.array_equal(scenes.dambreak_positions(n3, dx), po.dambreak_positions(n3, dx))


def test_reference_params_match_between_product_and_oracle():
    """dsl_params_reference (product, C ABI) and dslo_params_reference (oracle) agree."""
    from dieselfluid_amd.engine import reference_params
    for n3 in (4, 16, 20):
        p, q = reference_params(n3), po.params_reference(n3)
        for name in ("h", "mass", "ref_density", "mu", "dt", "eos_w", "eos_gamma", "eos_d0_grad", "pressure_sign",
                     "visc_running_mass", "pci_max_iters", "pci_max_error"):
            assert getattr(p, name) == getattr(q, name), name
        assert list(p.force_reset) == list(q.force_reset) and list(p.external) == list(q.external)
        assert p.n_particles == n3 ** 3 and p.lsh_bucket_size == po.lib().dslo_lsh_size(n3 ** 3, 255)
