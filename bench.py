#!/usr/bin/env python3
"""bench.py -- M particle-steps/s of the WCSPH dam-break step on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line
from rank 0.  A "step" is one full WCSPH step (neighbour-table build, density pass, fused
pressure+viscosity force + integrate) of the synthetic dam-break with the particle state
already resident in HBM.  N=1 workload: BASELINE.json configs[3]'s 16M-particle block
(n3=252 -> 16,003,008 particles) on one GPU; N>1: the same total problem partitioned into
spatial slabs (strong scaling), see dieselfluid_amd/slab.py.

Extra objects on the JSON line:
  roofline     -- dominant kernel: algorithmic bytes per launch / HIP-event duration
  cpu_baseline -- the CPU oracle (a port of the reference's single-threaded Go path)
                  timed on a bounded sample of the same workload, rank 0, N=1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic bytes per particle per launch (DESIGN.md "Kernels"):
#   density: read x (12) write rho (4)                       [SURVEY 8d: 16 B]
#   force+integrate: read x,v,rho (28), write x,v (24)        [SURVEY 8d: 52 B]
BYTES_DENSITY = 16
BYTES_FORCE = 52


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n3", type=int, default=252, help="fluid block edge in particles (252 -> 16.0M)")
    ap.add_argument("--math", choices=["fast", "exact"], default="fast")
    ap.add_argument("--method", choices=["wcsph", "pcisph"], default="wcsph",
                    help="pcisph: BASELINE configs[2] style run (use --n3 160 for 4.1M particles); not the bench line")
    ap.add_argument("--pci-iters", type=int, default=4)
    ap.add_argument("--extra-terms", action="store_true",
                    help="BASELINE configs[4]: add the build-defined XSPH + cohesion (surface tension) terms")
    ap.add_argument("--skin", type=float, default=None,
                    help="N=1 WCSPH, --math fast: DSL_OPT_SKIN, neighbour lists against h (1 + skin) that live until some "
                         "particle has moved skin h / 2 (0 = sort and sweep every step; default: the library's own default)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="a library option for A/B runs (include/dslsph.h DSL_OPT_*, e.g. pci_qincr=0); N=1 only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: exchange after the whole force pass")
    ap.add_argument("--developed-steps", type=int, default=10000,
                    help="N=1 WCSPH only: after the timed region advance this many steps (the column collapses, the "
                         "lattice melts: ~10000 steps = 0.45 s of flow at n3=252) and time 20 more; reported as "
                         "developed_ms_per_step beside the headline.  0 = skip")
    ap.add_argument("--drift-steps", type=int, default=400,
                    help="N=1 --method pcisph only: after the timed region advance this many steps -- the reference never "
                         "re-synchronises its predictor (pcisph_darwin.go:28-41), so DensityF's query points leave their "
                         "particles and the library sorts them into cells of their own (dsl_pcisph_set_binning) -- and time "
                         "20 more; reported under `drifted` beside the line.  0 = skip")
    ap.add_argument("--exact-steps", type=int, default=5,
                    help="N=1 WCSPH, --math fast only: also time this many steps of the SAME scene in DSL_MATH_EXACT (the "
                         "mode that is bit for bit the oracle's) on a second engine; reported under `exact`.  0 = skip")
    ap.add_argument("--cpu-n3", type=int, default=100,
                    help="edge of the CPU-baseline sample block (100 = BASELINE configs[1]'s 1M particles)")
    ap.add_argument("--cpu-steps", type=int, default=1)
    return ap.parse_args()


def cpu_baseline(args):
    """Oracle (single-threaded C port of the reference loops) on a bounded sample: the same
    dam-break scene and parameters at a smaller block, a few steps."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from dieselfluid_amd import scenes
    from oracle import pyoracle as po

    n3 = args.cpu_n3
    p, pos = scenes.dambreak_scene(n3)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (n3 ** 3, 1))
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
    t0 = time.perf_counter()
    ora.wcsph_step(args.cpu_steps)
    dt = time.perf_counter() - t0
    return {
        "value": round(n3 ** 3 * args.cpu_steps / dt / 1e6, 4),
        "unit": "M particle-steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"same dam-break scene at n3={n3} ({n3**3} particles), {args.cpu_steps} WCSPH step(s), "
                  f"{dt:.1f} s of CPU; oracle/dsl_oracle.c single thread",
    }


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, as a CHILD
    `torch.distributed.run` (this process has not touched the GPU and never will: it only waits for the child and
    exits with its code -- no exec from a process that has initialised HIP).  Under a launcher (WORLD_SIZE set) the
    world size must be the one asked for: a mismatch exits 2 instead of silently measuring another job."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(world_env or "1")
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         "(or without a launcher: bench.py starts the ranks itself)\n")
        raise SystemExit(2)
    return world


class Watchdog:
    """A collective that never completes (a rank missing, a group left open) must end the run with a non-zero exit
    code, not hang it: a timer thread ends the process -- os._exit, nothing is re-executed -- when a phase overruns.
    The main thread may be inside a C call (ctypes releases the GIL), the timer still fires."""

    def __init__(self, rank):
        self.rank = rank
        self.timer = None

    def arm(self, seconds, what):
        import threading
        self.disarm()

        def fire():
            sys.stderr.write(f"bench.py: rank {self.rank}: '{what}' did not finish within {seconds:.0f} s -- "
                             "a collective is stuck; exiting 4\n")
            sys.stderr.flush()
            os._exit(4)

        self.timer = threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def main():
    args = parse()
    world = launch_ranks(args)  # (before anything touches the GPU)
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("DSL_BENCH_BACKEND", "nccl")
    if backend != "nccl":  # rehearsal of the multi-rank path on a one-GPU box
        local_rank = int(os.environ.get("DSL_BENCH_DEVICE", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dog = Watchdog(rank)
    limit = float(os.environ.get("DSL_BENCH_WATCHDOG_S", "300"))
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dog.arm(limit, "process group set-up")
        # RCCL over xGMI; DSL_BENCH_BACKEND=gloo only exists to rehearse this code path with
        # several ranks on a one-GPU box (every rank then uses cuda:DSL_BENCH_DEVICE)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=limit))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=limit))
        if dist.get_world_size() != args.gpus:
            sys.stderr.write(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus asked for {args.gpus}\n")
            os._exit(2)

    from dieselfluid_amd import SPHEngine, scenes

    math_mode = 1 if args.math == "fast" else 0
    n3 = args.n3
    n_total = n3 ** 3

    if world == 1:
        p, pos = scenes.dambreak_scene(n3, math_mode=math_mode)
        if args.method == "pcisph":
            # the reference adds the EOS pressure gradient once per correction iteration
            # (pcisph_darwin.go:93): keep the total impulse of the WCSPH scene
            p.pci_max_iters = args.pci_iters
            p.eos_w = p.eos_w / args.pci_iters
            p.delta = 1.0e-7
        if args.extra_terms:
            p.xsph_eps = 0.25
            p.st_kappa = 25.0 * p.h * p.h  # cohesion acceleration ~ kappa/h^2: keep it resolution independent
        eng = SPHEngine(p, device=local_rank)
        eng.upload("positions", pos)
        eng.reset_forces()
        if args.skin is not None:
            eng.set_option("skin", args.skin)
        for kv in args.opt:
            eng.set_option(kv.split("=")[0], float(kv.split("=")[1]))
        del pos
        if args.method == "pcisph":
            eng.pcisph_begin()
            step = eng.pcisph_step
        else:
            step = eng.wcsph_step
        engines = [eng]
    else:
        from dieselfluid_amd import slab
        def pci_params(p):
            p.pci_max_iters = args.pci_iters
            p.eos_w = p.eos_w / args.pci_iters
            p.delta = 1.0e-7
            if args.extra_terms:
                p.xsph_eps = 0.25
                p.st_kappa = 25.0 * p.h * p.h

        pci = args.method == "pcisph"
        # engine kernels, RCCL calls and torch's own small ops are ordered through torch's current
        # stream: give it a stream of its own instead of the legacy default stream
        torch.cuda.set_stream(torch.cuda.Stream(torch.device("cuda", local_rank)))
        # the library drives the slab step (RCCL inside libdslsph.so); DSL_BENCH_SLAB_DRIVER=python keeps the
        # torch.distributed protocol of slab.py, =native forces the library driver also in a gloo rehearsal
        # (its transport calls then go through the host-staged table, engine.HostStagedComm)
        which = os.environ.get("DSL_BENCH_SLAB_DRIVER", "")
        drv = slab.SlabDriver.dambreak(n3, math_mode=math_mode, device=local_rank, pcisph=pci,
                                       params_hook=pci_params if pci else None,
                                       overlap=False if args.no_overlap else None,
                                       native={"python": False, "native": True}.get(which))
        step = drv.pcisph_step if pci else drv.wcsph_step
        engines = [drv.engine_core]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Per-kernel HIP events bracket launches: around every kernel they cost 1.8 % of the 2.5 ms step
    # at N=1 and about 10 % of a slab rank's 0.5 ms step.  N=1 therefore times only the two dominant
    # kernels (the roofline's) inside the timed region and the rest in a short segment after it;
    # N>1 times everything in that segment, `value` is measured without any.
    events_in_region = world == 1
    if world > 1:
        dog.arm(limit, "warm-up steps (first halo exchange)")
    step(args.warmup)
    for e in engines:
        e.timing_reset()
        e.timing_enable(2 if events_in_region else 0)
    barrier()
    if world > 1:
        dog.arm(limit, "timed steps")
    skin0 = (engines[0].get_option("skin_steps"), engines[0].get_option("skin_rebuilds")) if world == 1 else (0, 0)
    t0 = time.perf_counter()
    step(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    skin1 = (engines[0].get_option("skin_steps"), engines[0].get_option("skin_rebuilds")) if world == 1 else (0, 0)
    skin_fields = (engines[0].get_option("skin_fields_own"), engines[0].get_option("skin_fields_padded")) if world == 1 else (0, 0)
    def opt_or(name, default=0.0):  # (an older build loaded through DSL_LIB for an A/B run may not know the option)
        try:
            return engines[0].get_option(name)
        except Exception:
            return default
    skin_tau = opt_or("skin_tau_steps") if world == 1 else 0.0
    if world > 1:
        dog.arm(limit, "per-kernel timing segment and the closing reductions")
    hot = {k: engines[0].timing(k) for k in ("density", "force_integrate", "pci_density")}
    timed_launch_steps = min(args.steps, 10)
    for e in engines:
        e.timing_reset()
        e.timing_enable(True)
    step(timed_launch_steps)
    barrier()
    for e in engines:
        e.timing_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    eng = engines[0]
    overflow = band_missed = query_escaped = 0
    # how many ranks the communicator the halo exchange runs on REALLY spans (ncclCommCount inside the library;
    # the python protocol: the process group's own size) -- must be --gpus
    ranks_seen = 1
    if world > 1:
        nc = getattr(drv, "native_comm", None)
        ranks_seen = nc.count() if nc is not None else dist.get_world_size()
        if ranks_seen != args.gpus:
            sys.stderr.write(f"bench.py: the halo communicator spans {ranks_seen} ranks, --gpus asked for {args.gpus}\n")
            os._exit(2)
    if world > 1:
        # a band / capacity overflow or an outrun split margin would silently lose ghosts: make it visible
        # (third word: a PCISPH query point has drifted out of its rank's ghost coverage -- include/dslsph.h)
        st4 = eng.slab_status()
        ov = torch.tensor([st4[0], st4[1], int(eng.pcisph_query_escaped()) if args.method == "pcisph" else 0], dtype=torch.int64,
                          device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ov, op=dist.ReduceOp.MAX)
        overflow, band_missed, query_escaped = int(ov[0].item()), int(ov[1].item()), int(ov[2].item())
    def timing_of(k):  # N=1: the dominant kernels as timed inside the timed region
        return hot[k] if events_in_region and k in hot else eng.timing(k)

    ms_d, n_d = timing_of("density")
    ms_f, n_f = timing_of("force_integrate")
    steps_f = args.steps if events_in_region else timed_launch_steps
    if n_f > steps_f:  # split force pass: two launches per step, quote the pass
        ms_f = ms_f * n_f / steps_f
    # particles this rank integrates (ghosts take part in the sums but are not integrated)
    n_local = eng.n if world == 1 else eng.n_owned()
    n_live = eng.n
    if overflow or band_missed:
        sys.stderr.write(f"bench.py: slab exchange overflow={overflow} band_missed={band_missed}: particles or ghosts "
                         "were lost, the measurement is void\n")
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(3)
    if args.method == "pcisph":
        ms_f = timing_of("pci_density")[0]  # dominant PCISPH kernel: predicted density, 20 B/particle
        kname, kms, kbytes = "k_pci_density", ms_f, 20
    elif ms_f >= ms_d:
        kname, kms, kbytes = "k_force_integrate", ms_f, BYTES_FORCE
    else:
        kname, kms, kbytes = "k_density", ms_d, BYTES_DENSITY
    if world == 1 and args.method == "wcsph" and skin1[0] > skin0[0]:  # the timed steps walked neighbour lists
        kname = {"k_force_integrate": "k_force_list", "k_density": "k_density_list"}[kname]
    achieved = n_local * kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    st = eng.stats()
    kernels_ms = {k: round(timing_of(k)[0], 4) for k in
                  (("cell_rank", "scan", "scatter", "tile_list", "neigh_lists", "density", "force_integrate")
                   if args.method == "wcsph" else
                   ("cell_rank", "scan", "scatter", "tile_list", "density", "viscous", "gradient",
                    "pci_predict", "pci_density", "update"))}
    exact = None
    if world == 1 and args.method == "wcsph" and args.math == "fast" and args.exact_steps > 0:
        # the same scene on a second engine in DSL_MATH_EXACT: the reference's own float32 operations in the
        # reference's order (every EXACT parity test is array_equal against the oracle), driver-reproducible
        for e in engines:
            e.timing_enable(False)
        pe, pos_e = scenes.dambreak_scene(n3, math_mode=0)
        xe = SPHEngine(pe, device=local_rank)
        xe.upload("positions", pos_e)
        xe.reset_forces()
        del pos_e
        xe.wcsph_step(2)
        xe.timing_reset()
        xe.timing_enable(2)
        torch.cuda.synchronize()
        te = time.perf_counter()
        xe.wcsph_step(args.exact_steps)
        torch.cuda.synchronize()
        te = (time.perf_counter() - te) / args.exact_steps
        exact = {"value": round(n_total / te / 1e6, 3), "ms_per_step": round(te * 1e3, 4), "steps": args.exact_steps,
                 "warmup": 2, "kernels_ms": {k: round(xe.timing(k)[0], 4) for k in ("density", "force_integrate")},
                 "math": "exact (bit for bit the CPU oracle's results; tests/test_gpu_parity.py)"}
        xe.close()
        del xe
    developed = None
    if world == 1 and args.method == "wcsph" and args.developed_steps > 0:
        # the headline above is the contract's configuration (the jittered lattice: exactly 8 per cell);
        # this is the same engine once the flow has developed
        for e in engines:
            e.timing_enable(False)
        chunk = 500
        for _ in range(args.developed_steps // chunk):
            step(chunk)
        torch.cuda.synchronize()
        td = time.perf_counter()
        step(20)
        torch.cuda.synchronize()
        td = (time.perf_counter() - td) / 20
        eng.timing_reset()
        eng.timing_enable(True)
        step(5)
        dk = {k: round(eng.timing(k)[0], 4) for k in ("cell_rank", "scan", "scatter", "tile_list", "neigh_lists", "density", "force_integrate")}
        eng.timing_enable(False)
        sd = eng.stats()
        developed = {"ms_per_step": round(td * 1e3, 4), "value": round(n_total / td / 1e6, 3), "kernels_ms": dk,
                     "after_steps": args.warmup + args.steps + timed_launch_steps + (args.developed_steps // chunk) * chunk,
                     "max_cell_count": sd.max_cell_count, "max_vel": sd.max_vel}
    drifted = None
    if world == 1 and args.method == "pcisph" and args.drift_steps > 0:
        for e in engines:
            e.timing_enable(False)
        step(args.drift_steps)
        torch.cuda.synchronize()
        td = time.perf_counter()
        step(20)
        torch.cuda.synchronize()
        td = (time.perf_counter() - td) / 20
        eng.timing_reset()
        eng.timing_enable(True)
        step(5)
        dk = {k: round(eng.timing(k)[0], 4) for k in ("cell_rank", "scan", "scatter", "tile_list", "density", "viscous",
                                                      "pci_predict", "pci_density", "update")}
        eng.timing_enable(False)
        drifted = {"ms_per_step": round(td * 1e3, 4), "value": round(n_total / td / 1e6, 3), "kernels_ms": dk,
                   "after_steps": args.warmup + args.steps + timed_launch_steps + args.drift_steps,
                   "queries_binned": bool(eng.pcisph_binning()[1]),
                   "note": "pci_predict = predict + the queries' counting sort + query-tile tables, pci_density = the sweep; "
                           "per correction iteration"}
    # HBM bytes per launch and the vector-ALU counters cannot be sampled from inside this process; they come from the
    # committed rocprofv3 --pmc passes of this same command (profiles/traffic.json, written by tools/make_traffic.py from
    # the round's final profile) and are only quoted for the configuration and the kernel they were measured on.
    traffic = valu = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if world == 1 and tj.get("particles") == n_total and args.math == "fast" and kname in tj:
            traffic = tj[kname]["bytes"]
            valu = {"insts_per_launch": tj[kname]["insts_valu"], "issue_frac": tj[kname]["valu_issue_frac"],
                    "busy_frac": tj[kname]["valu_busy_frac"],
                    "definition": "issue_frac = SQ_INSTS_VALU x 2 / (1024 SIMDs x clocks), busy_frac = SQ_ACTIVE_INST_VALU x 4 / "
                                  "(1024 SIMDs x clocks); " + tj.get("source", "")}
    except Exception:
        traffic = valu = None
    # the density + force PASS in real traffic: both kernels' counter bytes over both kernels' time
    pass_traffic_frac = None
    try:
        kd, kf = ("k_density_list", "k_force_list") if kname.endswith("_list") else ("k_density", "k_force_integrate")
        if traffic is not None and kd in tj and kf in tj and (ms_d + ms_f) > 0:
            pass_traffic_frac = (tj[kd]["bytes"] + tj[kf]["bytes"]) / ((ms_d + ms_f) * 1e-3) / 1e9 / HBM_PEAK_GBS
    except Exception:
        pass_traffic_frac = None
    hbm_frac = achieved / HBM_PEAK_GBS
    # the roof that binds: the kernel's real HBM traffic against the peak, or the vector ALUs' busy share
    real_hbm_frac = (traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and kms > 0) else None
    bound = "hbm"
    if valu is not None and valu["busy_frac"] > max(hbm_frac, real_hbm_frac or 0.0):
        bound = "valu"

    if rank == 0:
        value = n_total * args.steps / dt / 1e6
        out = {
            # (the contract's line is the default run; other --method / --n3 are side measurements and say so in `config`)
            "metric": "M particle-steps/sec, 16M-particle WCSPH dam-break; % HBM roofline" if args.method == "wcsph"
            else "M particle-steps/sec, PCISPH dam-break (side measurement, see config); % HBM roofline",
            "value": round(value, 3),
            "unit": "M particle-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"WCSPH dam-break, {n_total} particles (n3={n3}), h=2dx, uniform-grid neighbours, "
                             f"density + pressure/viscosity force + integrate + walls, math={args.math}")
                if args.method == "wcsph" else
                (f"PCISPH dam-break ({args.pci_iters} correction iterations), {n_total} particles (n3={n3}), h=2dx, "
                 f"uniform-grid neighbours, math={args.math}") + (" + XSPH + cohesion terms" if args.extra_terms else ""),
                "particles": n_total,
                "parallelism": "single GPU" if world == 1 else f"{world} spatial slabs + 2h halo",
                **({"library_options": args.opt} if args.opt else {}),
            },
            "roofline": {
                "bound": bound,
                "kernel": kname,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_ratio": round(traffic / (n_local * kbytes), 3) if traffic else None,
                "traffic_frac_of_peak": round(real_hbm_frac, 4) if real_hbm_frac else None,
                "valu": valu,
                "avg_ms": round(kms, 4),
                "bytes_per_particle": kbytes,
                "pass_density_ms": round(ms_d, 4),
                "pass_force_ms": round(ms_f, 4),
                "pass_frac_68B": round(n_local * 68 / ((ms_d + ms_f) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                if (ms_d + ms_f) > 0 else None,
                # the same pass in the bytes the counters saw (lists, halo re-staging included): how busy HBM really is
                "pass_traffic_frac_of_peak": round(pass_traffic_frac, 4) if pass_traffic_frac else None,
            },
            "kernels_ms": kernels_ms,
            "slab_overflow": overflow,
            "slab_band_missed": band_missed,
            "slab_pci_query_escaped": query_escaped,
            "slab_overlap": bool(world > 1 and drv.overlap),
            "kernel_events": ("density / force kernels in the timed region, the others in a "
                              f"{timed_launch_steps}-step segment after it") if events_in_region else
                             f"separate {timed_launch_steps}-step segment after the timed region",
            "max_vel": st.max_vel,
            "max_cell_count": st.max_cell_count,
            "developed": developed,
            "drifted": drifted,
            "exact": exact,
            # DSL_OPT_SKIN: steps of the timed region that walked neighbour lists, and how many of them rebuilt the lists
            # (sort + candidate sweep) -- the region holds the share of rebuilds this phase of the flow asks for
            "skin": {"s": eng.get_option("skin"), "predict": opt_or("skin_predict"), "tau_steps_last_rebuild": round(skin_tau, 2),
                     "steps_in_timed_region": int(skin1[0] - skin0[0]),
                     "rebuilds_in_timed_region": int(skin1[1] - skin0[1]),
                     "list_overflow": int(eng.get_option("skin_list_overflow")),
                     # list fields per particle at the last rebuild: what the particles need, and what the walks visit
                     # once every wave's lists are padded to its longest
                     "fields_per_particle": round(skin_fields[0] / n_total, 2), "fields_walked_per_particle": round(skin_fields[1] / n_total, 2),
                     "suspensions": int(eng.get_option("skin_suspensions"))} if world == 1 else None,
            "n_live_rank0": n_live,
            "n_ranks_seen_by_rccl": ranks_seen,
            "slab_driver": (("native (dsl_slab_wcsph_step: RCCL inside libdslsph.so)" if backend == "nccl" else
                             "native (dsl_slab_wcsph_step over a host-staged dsl_comm_create_custom transport)")
                            if world > 1 and getattr(drv, "native", False)
                            else ("python protocol" if world > 1 else None)),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    dog.disarm()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
