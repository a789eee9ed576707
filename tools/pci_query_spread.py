"""How far apart do the DensityF queries of neighbouring particles land (tools only)?  For a saved drifted state: the number
of distinct query CELLS and query TILES among 64 / 1024 consecutive particles of the cell-sorted order."""
import sys
import numpy as np
sys.path.insert(0, ".")
from dieselfluid_amd import scenes

z = np.load(sys.argv[1])
n3 = int(z["n3"])
p, _ = scenes.dambreak_scene(n3, math_mode=1, positions=False)
h = np.float32(p.h)
g0 = np.array(p.grid_min[:], dtype=np.float32)
dims = np.ceil((np.array(p.grid_max[:], dtype=np.float32) - g0) / h).astype(np.int64)


def cells(a):
    c = np.floor((a - g0) / h).astype(np.int64)
    return np.clip(c, 0, dims - 1)


def lin(c, d):
    return (c[:, 2] * d[1] + c[:, 1]) * d[0] + c[:, 0]


x, xp = z["positions"], z["pci_positions"]
pc = cells(x)
order = np.argsort(lin(pc, dims), kind="stable")
qc = cells(xp)[order]
qcell = lin(qc, dims)
tdims = (dims + 3) // 4
qtile = lin(qc // 4, tdims)
for group in (64, 256, 1024):
    n = (qcell.shape[0] // group) * group
    a = np.sort(qcell[:n].reshape(-1, group), axis=1)
    b = np.sort(qtile[:n].reshape(-1, group), axis=1)
    dc = (np.diff(a, axis=1) != 0).sum(axis=1) + 1
    dt = (np.diff(b, axis=1) != 0).sum(axis=1) + 1
    print(f"group {group}: distinct query cells mean {dc.mean():.1f} (max {dc.max()}), distinct query tiles mean {dt.mean():.1f} (max {dt.max()})")
