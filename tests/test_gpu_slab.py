"""GPU: two slab ranks (both on cuda:0, transport gloo) drive the HIP engine through the
dsl_slab_* C ABI; the assembled result must match a single-engine run of the same scene.
The RCCL transport itself cannot be exercised on a one-GPU box; the driver code path is
identical except for the tensors' device."""
import functools
import os

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from oracle import pyoracle as po
from test_slab_cpu import _free_port, _vel_fn

pytestmark = pytest.mark.gpu

STEPS = int(os.environ.get("DSL_SLAB_TEST_STEPS", "10"))  # (the variable: tools/slab_exact_diff.py looks for the first differing step)


def _worker(rank, world, port, math_mode, n3, overlap, vscale, out, axis=2, native=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    torch.cuda.set_device(0)
    from dieselfluid_amd.slab import SlabDriver
    SlabDriver.REPLAN_EVERY = 4  # exercise the message re-sizing inside the run
    drv = SlabDriver.dambreak(n3, math_mode=math_mode, device=0, axis=axis, overlap=overlap, native=native,
                              vel_fn=lambda ids, pos: vscale * _vel_fn(ids, pos, axis=axis))
    assert drv.overlap == (overlap if overlap is not None else math_mode == 1)
    assert bool(getattr(drv, "native", False)) == native
    caps0 = (drv.engine.cap_full, drv.engine.cap_x)
    drv.wcsph_step(STEPS)
    res = drv.gather_state(n3 ** 3)
    st = drv.engine.status()
    info = [None] * world if rank == 0 else None
    dist.gather_object((drv.engine.n, drv.engine_core.n_owned(), st[0], st[1], caps0[0], drv.engine.cap_full), info, dst=0)
    calls = [None] * world if rank == 0 else None
    dist.gather_object(list(drv.engine_core._comm.calls) if native else [], calls, dst=0)
    if rank == 0:
        gp, gv, seen = res
        np.savez(out, pos=gp, vel=gv, seen=seen, info=np.array(info))
        if native:
            import pickle
            with open(out + ".calls", "wb") as f:
                pickle.dump(calls, f)
    dist.destroy_process_group()


def _single(n3, math_mode, vscale, steps=STEPS, shuffle=False, axis=2):
    """single-engine run; `shuffle` also re-orders the particles and moves the grid origin (other
    cells, other tiles, other tile-relative roundings in FAST mode): what a slab rank's own grid does"""
    from dieselfluid_amd import SPHEngine, scenes
    p, pos = scenes.dambreak_scene(n3, math_mode=math_mode)
    if shuffle:
        for a in range(3):
            p.grid_min[a] -= 0.37 * p.h
    vel = vscale * _vel_fn(np.arange(n3 ** 3), pos, axis)
    perm = np.random.default_rng(0).permutation(n3 ** 3) if shuffle else np.arange(n3 ** 3)
    eng = SPHEngine(p)
    eng.upload("positions", pos[perm])
    eng.upload("velocities", vel[perm])
    eng.reset_forces()
    eng.wcsph_step(steps)
    inv = np.argsort(perm)
    return eng.download("positions")[inv], eng.download("velocities")[inv]


def _bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                          np.ascontiguousarray(b, np.float32).view(np.uint32))


def _oracle_single(n3, vscale, steps=STEPS, axis=2, pcisph=False, params_hook=None):
    """the same scene on the CPU oracle, single domain, EXACT arithmetic in the reference's order"""
    from dieselfluid_amd import scenes
    p, pos = scenes.dambreak_scene(n3, math_mode=0)
    if params_hook is not None:
        params_hook(p)
    vel = (vscale * _vel_fn(np.arange(n3 ** 3), pos, axis)).astype(np.float32)
    frc = np.tile(np.array(p.force_reset[:], np.float32), (n3 ** 3, 1))
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=vel, force=frc)
    if pcisph:
        ora.delta = p.delta
        ora.pcisph_begin()
        ora.pcisph_step(steps)
    else:
        ora.wcsph_step(steps)
    return ora.positions(), ora.velocities()


def _assert_identical(what, got_pos, got_vel, pos, vel):
    """DSL_MATH_EXACT: cells are ascending in (global) particle id, slab planes and every rank's grid origin lie on
    cell planes of the single-domain grid, ghost densities come from a complete 2h band -- every sum of every owned
    particle has the same terms in the same order as in the single-domain run, so the states are the same BITS."""
    nx = int(np.count_nonzero(np.ascontiguousarray(got_pos, np.float32).view(np.uint32) != pos.view(np.uint32)))
    nv = int(np.count_nonzero(np.ascontiguousarray(got_vel, np.float32).view(np.uint32) != vel.view(np.uint32)))
    assert nx == 0 and nv == 0, f"EXACT slabs differ from {what}: {nx} position words, {nv} velocity words"


@pytest.mark.parametrize("math_mode,overlap,world,n3", [
    (0, None, 2, 16),    # EXACT kernels, exchange after the force pass
    (1, False, 2, 16),   # tiled kernels, exchange after the force pass
    (1, True, 2, 16),    # split force pass, every tile is a band tile
    (1, True, 2, 32),    # split force pass with interior tiles
    (0, None, 3, 24),    # a middle rank with two neighbours
    (1, False, 3, 24),
    (1, True, 3, 24),
])
def test_hip_slabs_match_single_engine(tmp_path, math_mode, overlap, world, n3):
    out = str(tmp_path / "slab_gpu.npz")
    mp.spawn(_worker, args=(world, _free_port(), math_mode, n3, overlap, 1.0, out), nprocs=world, join=True)
    z = np.load(out)
    info = z["info"]
    assert np.all(z["seen"] == 1)
    assert info[:, 0].sum() > n3 ** 3       # ghosts are present
    assert info[:, 1].sum() == n3 ** 3      # every particle owned exactly once
    assert np.all(info[:, 2] == 0)          # no message / capacity overflow
    assert np.all(info[:, 3] == 0)          # split step: the margin held
    assert np.all(info[:, 5] != info[:, 4])  # messages were re-sized to the band occupancy
    pos, vel = _single(n3, math_mode, 1.0)
    if math_mode == 0:
        # EXACT: identical, not close -- to the single engine and to the single-domain oracle
        _assert_identical("the single engine", z["pos"], z["vel"], pos, vel)
        opos, ovel = _oracle_single(n3, 1.0)
        _assert_identical("the single-domain oracle", z["pos"], z["vel"], opos, ovel)
        return
    # FAST: tile-relative coordinates round differently on a rank's own grid.  The two counter-streaming
    # particle populations of this scene amplify float32 rounding noise quickly; the yardstick is what
    # re-ordering the particles and moving the grid origin does to a single-engine run.  Slabs (own order
    # and own grid per rank) must stay within a small multiple.
    pos_s, vel_s = _single(n3, math_mode, 1.0, shuffle=True)
    tol_x = max(4e-6, 5.0 * helpers.rel_err(pos_s, pos))
    tol_v = max(1e-4, 5.0 * helpers.rel_err(vel_s, vel))
    ex, ev = helpers.rel_err(z["pos"], pos), helpers.rel_err(z["vel"], vel)
    print(f"slab-vs-single x {ex:.2e} (tol {tol_x:.2e})  v {ev:.2e} (tol {tol_v:.2e})")
    assert ex < tol_x
    assert ev < tol_v


@pytest.mark.parametrize("axis", [0, 1])
def test_hip_slabs_along_x_and_y(tmp_path, axis):
    """the slab axis is a parameter: rows run along x, so x slabs cut through rows and y slabs between them"""
    n3, world, math_mode = 32, 2, 1
    out = str(tmp_path / "slab_gpu.npz")
    mp.spawn(_worker, args=(world, _free_port(), math_mode, n3, True, 1.0, out, axis), nprocs=world, join=True)
    z = np.load(out)
    info = z["info"]
    assert np.all(z["seen"] == 1)
    assert info[:, 1].sum() == n3 ** 3
    assert np.all(info[:, 2] == 0) and np.all(info[:, 3] == 0)
    pos, vel = _single(n3, math_mode, 1.0, axis=axis)
    pos_s, vel_s = _single(n3, math_mode, 1.0, shuffle=True, axis=axis)
    tol_x = max(4e-6, 5.0 * helpers.rel_err(pos_s, pos))
    tol_v = max(1e-4, 5.0 * helpers.rel_err(vel_s, vel))
    assert helpers.rel_err(z["pos"], pos) < tol_x
    assert helpers.rel_err(z["vel"], vel) < tol_v


def _violation_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    torch.cuda.set_device(0)
    from dieselfluid_amd.slab import SlabDriver, SlabOverflow
    SlabDriver.REPLAN_EVERY = 4
    drv = SlabDriver.dambreak(32, math_mode=1, device=0, axis=2, overlap=True,
                              vel_fn=lambda ids, pos: 8.0 * _vel_fn(ids, pos, axis=2))
    raised = False
    try:
        drv.wcsph_step(STEPS)
    except SlabOverflow as e:
        raised = "margin" in str(e)
    flags = [None] * world if rank == 0 else None
    dist.gather_object(raised, flags, dst=0)
    if rank == 0:
        np.savez(out, raised=np.array(flags))
    dist.destroy_process_group()


def test_split_step_stops_on_a_margin_violation(tmp_path):
    """particles faster than margin / dt outrun the band tiles: EVERY rank must stop with SlabOverflow at
    the next re-plan (the status words are MAX-reduced over the ranks), not carry on with ghosts missing"""
    out = str(tmp_path / "slab_gpu.npz")
    mp.spawn(_violation_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert np.all(np.load(out)["raised"])


# ---- PCISPH across slabs (BASELINE configs[4]'s 8-GPU form) ---------------------------------

def _pci_params(p, extra):
    p.pci_max_iters = 4
    p.eos_w = p.eos_w / 4          # as bench.py: the EOS gradient is added once per iteration
    p.delta = 1.0e-7
    p.pci_max_error = 1.0          # reached in the 16^3 cases (early-out after one iteration), not in the 24^3 one
    if extra:
        p.xsph_eps = 0.25
        p.st_kappa = 25.0 * p.h * p.h


def _pci_worker(rank, world, port, math_mode, n3, extra, out, native=False, steps=None, binned=0):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    torch.cuda.set_device(0)
    from dieselfluid_amd.slab import SlabDriver
    drv = SlabDriver.dambreak(n3, math_mode=math_mode, device=0, axis=2, pcisph=True, native=native,
                              params_hook=lambda p: _pci_params(p, extra),
                              vel_fn=lambda ids, pos: 0.5 * _vel_fn(ids, pos, axis=2))
    assert drv.engine.eng.slab_record_floats() == 13
    drv.engine_core.pcisph_set_binning(1 if binned else 0)
    drv.pcisph_step(STEPS if steps is None else steps)
    res = drv.gather_state(n3 ** 3)
    st = drv.engine.status()
    stats = drv.engine_core.stats()
    info = [None] * world if rank == 0 else None
    dist.gather_object((drv.engine_core.n_owned(), st[0], stats.pci_iters, stats.pci_max_error,
                        int(drv.engine_core.pcisph_query_escaped())), info, dst=0)
    calls = [None] * world if rank == 0 else None
    dist.gather_object(list(drv.engine_core._comm.calls) if native else [], calls, dst=0)
    if rank == 0:
        gp, gv, seen = res
        np.savez(out, pos=gp, vel=gv, seen=seen, info=np.array(info, dtype=np.float64))
        if native:
            import pickle
            with open(out + ".calls", "wb") as f:
                pickle.dump(calls, f)
    dist.destroy_process_group()


def _pci_single(n3, math_mode, extra, shuffle=False, binned=0):
    from dieselfluid_amd import SPHEngine, scenes
    p, pos = scenes.dambreak_scene(n3, math_mode=math_mode)
    _pci_params(p, extra)
    if shuffle:
        for a in range(3):
            p.grid_min[a] -= 0.37 * p.h
    vel = 0.5 * _vel_fn(np.arange(n3 ** 3), pos, 2)
    perm = np.random.default_rng(0).permutation(n3 ** 3) if shuffle else np.arange(n3 ** 3)
    eng = SPHEngine(p)
    eng.upload("positions", pos[perm])
    eng.upload("velocities", vel[perm])
    eng.reset_forces()
    eng.pcisph_begin()
    eng.pcisph_set_binning(1 if binned else 0)
    eng.pcisph_step(STEPS)
    inv = np.argsort(perm)
    st = eng.stats()
    return eng.download("positions")[inv], eng.download("velocities")[inv], st.pci_iters, st.pci_max_error


@pytest.mark.parametrize("math_mode,world,n3,extra,binned", [(0, 2, 16, False, 0), (0, 3, 24, False, 0), (0, 3, 24, True, 0),
                                                             (1, 2, 16, True, 0), (1, 3, 24, True, 0),
                                                             (0, 3, 24, True, 1), (1, 3, 24, True, 1)])
def test_pcisph_slabs_match_single_engine(tmp_path, monkeypatch, math_mode, world, n3, extra, binned):
    """binned = 1: every engine (the ranks' and the single one) sorts DensityF's query points into cells of their own
    (dsl_pcisph_set_binning(1)) -- ghosts are no queries there either"""
    out = str(tmp_path / "slab_pci.npz")
    mp.spawn(_pci_worker, args=(world, _free_port(), math_mode, n3, extra, out, False, None, binned), nprocs=world, join=True)
    z = np.load(out)
    info = z["info"]
    assert np.all(z["seen"] == 1)
    assert info[:, 0].sum() == n3 ** 3                      # every particle owned exactly once
    assert np.all(info[:, 1] == 0)                          # no message / capacity overflow
    assert len(set(info[:, 2].tolist())) == 1               # every rank ran the same number of iterations
    assert np.allclose(info[:, 3], info[0, 3], rtol=0, atol=0)  # ... on the same, global, error
    pos, vel, iters, err = _pci_single(n3, math_mode, extra, binned=binned)
    assert int(info[0, 2]) == iters
    if math_mode == 0:
        # EXACT: the same bits as the single engine (error word included) and as the single-domain oracle
        assert np.float32(info[0, 3]) == np.float32(err)
        _assert_identical("the single engine", z["pos"], z["vel"], pos, vel)
        opos, ovel = _oracle_single(n3, 0.5, pcisph=True, params_hook=lambda p: _pci_params(p, extra))
        _assert_identical("the single-domain oracle", z["pos"], z["vel"], opos, ovel)
        return
    assert abs(info[0, 3] - err) <= 1e-3 * max(err, 1e-6)
    pos_s, vel_s, _, _ = _pci_single(n3, math_mode, extra, shuffle=True, binned=binned)
    tol_x = max(4e-6, 5.0 * helpers.rel_err(pos_s, pos))
    tol_v = max(1e-4, 5.0 * helpers.rel_err(vel_s, vel))
    ex, ev = helpers.rel_err(z["pos"], pos), helpers.rel_err(z["vel"], vel)
    print(f"pcisph slab-vs-single x {ex:.2e} (tol {tol_x:.2e})  v {ev:.2e} (tol {tol_v:.2e})  iters {iters} err {err:.3e}")
    assert ex < tol_x
    assert ev < tol_v


def test_pcisph_slabs_report_queries_that_left_the_ghost_coverage(tmp_path):
    """The reference never re-synchronises its predictor, so DensityF's query points drift away from their particles; a
    slab's ghosts cover 2h beyond its planes, i.e. queries up to h beyond them.  A run stays the single-domain run only
    until the first query is further out -- the library says when (dsl_pcisph_get_binning's third word), here within 40
    steps of a 24^3 block on three ranks (slabs 4h thick), and not in the first."""
    out = str(tmp_path / "slab_pci_escape.npz")
    flags = []
    for steps in (1, 40):
        mp.spawn(_pci_worker, args=(3, _free_port(), 1, 24, False, out, False, steps), nprocs=3, join=True)
        info = np.load(out)["info"]
        assert np.all(info[:, 1] == 0)
        flags.append(int(info[:, 4].max()))
    assert flags == [0, 1]


def _load_calls(out):
    import pickle
    with open(out + ".calls", "rb") as f:
        return pickle.load(f)


def _check_exchange_calls(calls, world):
    """what the library asked of the transport, rank by rank: every exchange is one group of a send and a recv per
    neighbour (sends first, then the receives in lo, hi order), a middle rank talks to two DISTINCT peers, and the
    re-plan's 4-word all-reduce happened on every rank"""
    for rank, seq in enumerate(calls):
        nbs = [r for r in (rank - 1, rank + 1) if 0 <= r < world]
        groups, cur = [], None
        for c in seq:
            if c == "group_start":
                cur = []
            elif c == "group_end":
                groups.append(cur)
                cur = None
            elif isinstance(c, tuple) and c[0] in ("send", "recv"):
                assert cur is not None, "a transfer outside a group"
                cur.append(c)
        assert len(groups) >= STEPS
        for g in groups:
            assert [c[0] for c in g] == ["send"] * len(nbs) + ["recv"] * len(nbs)
            assert [c[1] for c in g] == nbs + nbs
            assert len({c[2] for c in g}) == 1  # one message size per exchange, the same in both directions
        assert ("all_reduce_max", 4) in seq


@pytest.mark.parametrize("math_mode,overlap,world,n3", [
    (0, None, 2, 16),    # EXACT, unsplit
    (1, True, 2, 32),    # split step, interior tiles
    (1, False, 3, 24),   # a middle rank: two distinct peers
    (1, True, 3, 24),
])
def test_native_step_driver_between_distinct_ranks(tmp_path, math_mode, overlap, world, n3):
    """the library's own step driver (dsl_slab_wcsph_step: group send/recv, split step,
    re-plan all-reduce) between 2 and 3 DISTINCT ranks.  RCCL refuses several ranks on one device, so the
    communicator is a dsl_comm_create_custom one whose table stages the messages through the host and gloo
    (engine.HostStagedComm) -- the library makes the same call sequence as to RCCL.  Bit for bit the Python
    protocol's result, in both math modes."""
    out_py, out_nat = str(tmp_path / "py.npz"), str(tmp_path / "native.npz")
    mp.spawn(_worker, args=(world, _free_port(), math_mode, n3, overlap, 1.0, out_py), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), math_mode, n3, overlap, 1.0, out_nat, 2, True), nprocs=world, join=True)
    a, b = np.load(out_py), np.load(out_nat)
    assert np.all(b["seen"] == 1)
    assert b["info"][:, 1].sum() == n3 ** 3
    assert np.all(b["info"][:, 2] == 0) and np.all(b["info"][:, 3] == 0)
    assert _bits_equal(a["pos"], b["pos"]) and _bits_equal(a["vel"], b["vel"])
    _check_exchange_calls(_load_calls(out_nat), world)


@pytest.mark.parametrize("math_mode,world,n3,extra", [(0, 2, 16, False), (1, 3, 24, True)])
def test_native_pcisph_driver_between_distinct_ranks(tmp_path, math_mode, world, n3, extra):
    """dsl_slab_pcisph_step between distinct ranks: one exchange per step plus one 1-word MAX all-reduce of the
    iteration error per correction iteration, on every rank; same bits as the Python protocol."""
    out_py, out_nat = str(tmp_path / "py.npz"), str(tmp_path / "native.npz")
    mp.spawn(_pci_worker, args=(world, _free_port(), math_mode, n3, extra, out_py), nprocs=world, join=True)
    mp.spawn(_pci_worker, args=(world, _free_port(), math_mode, n3, extra, out_nat, True), nprocs=world, join=True)
    a, b = np.load(out_py), np.load(out_nat)
    assert np.all(b["seen"] == 1)
    assert np.array_equal(a["info"], b["info"])  # owned counts, status, iteration count, error: identical
    assert _bits_equal(a["pos"], b["pos"]) and _bits_equal(a["vel"], b["vel"])
    calls = _load_calls(out_nat)
    for seq in calls:
        n_red = sum(1 for c in seq if c == ("all_reduce_max", 1))
        assert n_red == STEPS * 4, n_red  # pci_max_iters = 4: the reduce runs in every iteration, converged or not
        assert ("all_reduce_max", 4) in seq


def test_native_step_driver_matches_the_python_protocol():
    """The exchange behind the C ABI (dsl_slab_attach / dsl_slab_wcsph_step: RCCL group send/recv
    issued by the library, split step, re-plan) against the Python protocol it replaces, on a middle
    rank of a 4-way split whose neighbours are its own periodic images: the library sends both bands
    through an RCCL communicator of one rank to itself (dsl_slab_image_shift moves them by the slab
    thickness), the Python driver moves them by device copies.  Same kernels, same message contents,
    same (id-ordered) cells: the owned states must agree bit for bit."""
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from slab_periodic_bench import PeriodicDriver
    from dieselfluid_amd.engine import Comm

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        res = {}
        for kind in ("python", "native"):
            for overlap in (False, True):
                drv = PeriodicDriver.dambreak(64, math_mode=1, device=0, rank=1, world=4, overlap=overlap, native=False)
                drv.comm_dev, drv.use_nccl = torch.device("cpu"), False
                if kind != "python":
                    comm = Comm(1, 0, 0)
                    drv.attach_native(comm, 0, 0)
                    T = drv.hi - drv.lo
                    drv.engine_core.slab_image_shift(-T, +T)
                drv.wcsph_step(20)
                torch.cuda.synchronize()
                ids, pos, vel = drv.engine.owned_state(drv.axis, drv.lo, drv.hi)
                o = np.argsort(ids)
                st = drv.engine.status()
                assert st[0] == 0 and st[1] == 0
                res[(kind, overlap)] = (ids[o], pos[o], vel[o])
                drv.engine_core.close()
        for overlap in (False, True):
            for kind in ("native",):
                a, b = res[("python", overlap)], res[(kind, overlap)]
                assert np.array_equal(a[0], b[0]) and a[0].shape[0] > 60000
                assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
                assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    finally:
        if created:
            dist.destroy_process_group()


def test_native_driver_reports_an_overflow_instead_of_losing_particles():
    """a band that does not fit its message must stop the run.  Message capacities far
    below the band occupancy: the re-plan (every 8th step, or on demand) fails with DSL_ERR_OVERFLOW."""
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from slab_periodic_bench import PeriodicDriver
    from dieselfluid_amd import slab
    from dieselfluid_amd.engine import Comm

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29592")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        drv = PeriodicDriver.dambreak(32, math_mode=1, device=0, rank=1, world=4, overlap=False, native=False)
        drv.engine.cap_full, drv.engine.cap_x = 64, 64  # a band holds thousands
        drv.engine.max_full, drv.engine.max_x = 128, 128
        comm = Comm(1, 0, 0)
        drv.attach_native(comm, 0, 0)
        T = drv.hi - drv.lo
        drv.engine_core.slab_image_shift(-T, +T)
        with pytest.raises(slab.SlabOverflow):
            drv.wcsph_step(8)
        drv.engine_core.close()
        # the Python protocol raises as well (slab.py: _replan)
        drv = PeriodicDriver.dambreak(32, math_mode=1, device=0, rank=1, world=4, overlap=False, native=False)
        drv.comm_dev, drv.use_nccl = torch.device("cpu"), False
        drv.engine.cap_full, drv.engine.cap_x = 64, 64
        drv.engine.max_full, drv.engine.max_x = 128, 128
        with pytest.raises(slab.SlabOverflow):
            drv.wcsph_step(8)
        drv.engine_core.close()
    finally:
        if created:
            dist.destroy_process_group()
