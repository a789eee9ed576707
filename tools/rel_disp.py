#!/usr/bin/env python3
"""How long would neighbour lists live under a criterion on RELATIVE displacement?  (tools only; GPU box)

The skin step (DSL_OPT_SKIN) rebuilds when some particle has moved s h / 2 since the build -- an absolute bound that the
developed dam-break (bulk at 0.02-0.03 h per step, moving coherently) exhausts every step.  What the lists really need is
that no UNLISTED pair comes within h: |d_i - d_j| <= |x0_i - x0_j| - h, d = displacement since the build.  With the
particles grouped by their build cell (edge E = h (1 + s)), m_c = a cell's mean displacement and rho_c = max |d_i - m_c|:
  near (cells at most 2 apart, unlisted pairs start >= E apart):  rho_a + rho_b + |m_a - m_b| <= s h
  far  (cells >= 3 apart, pairs start >= 2 E apart):              |d_i - d_j| <= h (1 + 2 s)
This tool advances the bench scene to a given step, then measures both quantities after k = 1, 2, 3, ... further steps and
says how many cells would break the near bound, and where (walls, lone particles)."""
import argparse, json, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dieselfluid_amd import SPHEngine, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--n3", type=int, default=252)
ap.add_argument("--at", type=str, default="10000")
ap.add_argument("--ks", type=str, default="1,2,3,4,6,8,12,16,24,32")
ap.add_argument("--s", type=float, default=0.08)
a = ap.parse_args()
dev = torch.device("cuda:0")
p, pos = scenes.dambreak_scene(a.n3)
h, s = float(p.h), a.s
E = h * (1.0 + s)
gmin = np.array([p.grid_min[i] for i in range(3)], dtype=np.float64)
gmax = np.array([p.grid_max[i] for i in range(3)], dtype=np.float64)
dims = np.maximum(1, np.ceil((gmax - gmin) / E).astype(np.int64))
nx, ny, nz = (int(v) for v in dims)
eng = SPHEngine(p, device=0)
eng.upload("positions", pos); eng.reset_forces(); del pos
done = 0


def advance(to):
    global done
    while done < to:
        k = min(500, to - done); eng.wcsph_step(k); done += k


def analyse(x0, x1, k, step0):
    x0t = torch.from_numpy(x0).to(dev).double()
    d = (torch.from_numpy(x1).to(dev).double() - x0t) / h  # in units of h
    ci = torch.floor((x0t - torch.tensor(gmin, device=dev)) / E).long()
    for ax, n in enumerate((nx, ny, nz)):
        ci[:, ax].clamp_(0, n - 1)
    cell = (ci[:, 0] * ny + ci[:, 1]) * nz + ci[:, 2]
    nc = nx * ny * nz
    cnt = torch.zeros(nc, device=dev, dtype=torch.float64).index_add_(0, cell, torch.ones_like(d[:, 0]))
    m = torch.zeros(nc, 3, device=dev, dtype=torch.float64).index_add_(0, cell, d)
    m = m / cnt.clamp(min=1.0)[:, None]
    dev_i = (d - m[cell]).norm(dim=1)
    rho = torch.zeros(nc, device=dev, dtype=torch.float64).scatter_reduce_(0, cell, dev_i, "amax", include_self=True)
    ne = cnt > 0
    M = m.view(nx, ny, nz, 3); R = rho.view(nx, ny, nz); NE = ne.view(nx, ny, nz)
    crit = R.clone() * 2.0  # the cell with itself
    crit[~NE] = 0.0
    for ox in range(0, 3):
        for oy in range(-2, 3):
            for oz in range(-2, 3):
                if (ox, oy, oz) <= (0, 0, 0):
                    continue
                sa = (slice(0, nx - ox), slice(max(0, -oy), ny - max(0, oy)), slice(max(0, -oz), nz - max(0, oz)))
                sb = (slice(ox, nx), slice(max(0, oy), ny - max(0, -oy)), slice(max(0, oz), nz - max(0, -oz)))
                v = R[sa] + R[sb] + (M[sa] - M[sb]).norm(dim=-1)
                v = torch.where(NE[sa] & NE[sb], v, torch.zeros_like(v))
                crit[sa] = torch.maximum(crit[sa], v)
                crit[sb] = torch.maximum(crit[sb], v)
    cne = crit[NE]
    qs = [50, 90, 99, 99.9, 99.99]
    # torch.quantile has an input size limit: numpy on the host
    cn = cne.cpu().numpy()
    viol = (crit > s) & NE
    nviol = int(viol.sum())
    idx = viol.nonzero()
    near_wall = 0
    lone = 0
    if nviol:
        bmin = np.array([p.box_min[i] for i in range(3)]); bmax = np.array([p.box_max[i] for i in range(3)])
        lo = torch.tensor(np.floor((bmin - gmin) / E), device=dev); hi = torch.tensor(np.floor((bmax - gmin) / E), device=dev)
        nw = ((idx - lo).abs().min(dim=1).values <= 2) | ((idx - hi).abs().min(dim=1).values <= 2)
        near_wall = int(nw.sum())
        lone = int((cnt.view(nx, ny, nz)[viol] <= 2).sum())
    dabs = d.norm(dim=1)
    gm = d.mean(dim=0)
    out = {"at_step": step0, "k": k, "s": s, "cells_nonempty": int(ne.sum()),
           "abs_disp_over_h": {"max": float(dabs.max()), "99.9": float(np.percentile(dabs.cpu().numpy(), 99.9)), "mean": float(dabs.mean())},
           "abs_minus_global_mean_max": float((d - gm).norm(dim=1).max()),
           "within_cell_rho": {"max": float(rho.max()), "99.9": float(np.percentile(rho[ne].cpu().numpy(), 99.9)), "50": float(np.percentile(rho[ne].cpu().numpy(), 50))},
           "near_crit_over_h": {str(q): float(np.percentile(cn, q)) for q in qs} | {"max": float(cn.max())},
           "cells_over_budget": nviol, "of_them_within_2_cells_of_a_wall": near_wall, "of_them_with_at_most_2_particles": lone,
           "particles_in_cells_over_budget": int(cnt.view(nx, ny, nz)[viol].sum()) if nviol else 0}
    print(json.dumps(out), flush=True)


ks = [int(v) for v in a.ks.split(",")]
for at in [int(v) for v in a.at.split(",")]:
    advance(at)
    x0 = eng.download("positions")
    for k in ks:
        advance(at + k)
        x1 = eng.download("positions")
        analyse(x0, x1, k, at)
eng.close()
