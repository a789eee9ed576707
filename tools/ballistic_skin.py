#!/usr/bin/env python3
"""How long would the skin step's lists live if they were built at PREDICTED positions?  (tools only; GPU box)

A list is valid while every particle is within s h / 2 of the REFERENCE position its list was built at -- any reference
will do.  Today that is where the particle was at the build (x0); with x0' = x0 + tau v0 the same budget covers the
trajectory from -tau v0 to +tau v0 around the reference: twice the steps if particles move ballistically.  This tool
advances the bench scene to step T, keeps x(T), v(T), then steps on and prints, for several tau (in steps),
max_i |x_i(T + k) - x_i(T) - tau dt v_i(T)| / h for k = 0, 1, 2, ...: the number of steps each tau would have lasted
under a budget of s h / 2."""
import argparse, json, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dieselfluid_amd import SPHEngine, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--n3", type=int, default=252)
ap.add_argument("--at", type=str, default="20,60,500")
ap.add_argument("--kmax", type=int, default=64)
ap.add_argument("--taus", type=str, default="0,4,6,8,10,12,16,20")
ap.add_argument("--s", type=float, default=0.08)
ap.add_argument("--skin", type=float, default=-1.0, help="DSL_OPT_SKIN of the engine that advances the scene (-1: its default)")
a = ap.parse_args()
dev = torch.device("cuda:0")
p, pos = scenes.dambreak_scene(a.n3)
h, dt = float(p.h), float(p.dt)
budget = a.s / 2.0
eng = SPHEngine(p, device=0)
if a.skin >= 0.0:
    eng.set_option("skin", a.skin)
eng.upload("positions", pos); eng.reset_forces(); del pos
done = 0
taus = [float(v) for v in a.taus.split(",")]
for at in [int(v) for v in a.at.split(",")]:
    while done < at - 1:
        k = min(500, at - 1 - done); eng.wcsph_step(k); done += k
    vm1 = torch.from_numpy(eng.download("velocities")).to(dev).double()
    eng.wcsph_step(1); done += 1
    x0 = torch.from_numpy(eng.download("positions")).to(dev).double()
    v0 = torch.from_numpy(eng.download("velocities")).to(dev).double()
    a0 = (v0 - vm1) / dt  # the acceleration of the last step: a second-order reference x0 + tau v0 + tau^2 a0 / 2
    vmax = float(v0.norm(dim=1).max()) * dt / h
    table = {str(t): [] for t in taus}
    table2 = {str(t): [] for t in taus}
    for k in range(0, a.kmax + 1):
        if k > 0:
            eng.wcsph_step(1); done += 1
            x = torch.from_numpy(eng.download("positions")).to(dev).double()
        else:
            x = x0
        d = (x - x0) / h
        for t in taus:
            table[str(t)].append(float((d - (t * dt / h) * v0).norm(dim=1).max()))
            table2[str(t)].append(float((d - (t * dt / h) * v0 - (0.5 * (t * dt) ** 2 / h) * a0).norm(dim=1).max()))
    def lives(tab):
        life = {}
        for t in taus:
            row = tab[str(t)]
            n = 0
            while n < len(row) and row[n] <= budget:
                n += 1
            life[str(t)] = n - 1 if n > 0 else -1  # last k that was still inside the budget (kmax: never left it); -1: the reference itself is outside
        return life
    life = lives(table)
    print(json.dumps({"at_step": at, "s": a.s, "budget_over_h": budget, "max_v_dt_over_h": vmax, "steps_inside_budget": life,
                      "steps_inside_budget_second_order": lives(table2),
                      "max_dev_over_h": {t: [round(v, 5) for v in row[:: max(1, a.kmax // 16)]] for t, row in table.items()}}), flush=True)
eng.close()
