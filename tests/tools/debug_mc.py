"""GPU debug: locate the first cell where HIP marching cubes diverges from the oracle."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import capi
from sculptmate_amd import ops

def cell_tables(vol):
    L = capi.lib()
    n0, n1, n2 = vol.shape
    out = []
    for z in range(n0 - 1):
        for y in range(n1 - 1):
            for x in range(n2 - 1):
                c = vol.astype(np.float64)
                v = np.array([c[z, y, x], c[z, y, x+1], c[z, y+1, x+1], c[z, y+1, x], c[z+1, y, x], c[z+1, y, x+1], c[z+1, y+1, x+1], c[z+1, y+1, x]])
                t = ctypes.c_int(); r = ctypes.c_int(); s = ctypes.c_int()
                nt = L.oracle_mc_classify(v.ctypes.data, ctypes.byref(t), ctypes.byref(r), ctypes.byref(s))
                out.append(((z, y, x), nt, t.value, r.value, s.value, v))
    return out

name = sys.argv[1] if len(sys.argv) > 1 else "ints"
G = np.load("tests/golden/mc_skimage.npz")
vol = G[name + "_vol"]
rv, rf = capi.marching_cubes(vol, 0.0)
v, f = ops.marching_cubes(torch.from_numpy(vol).cuda(), 0.0)
f = f.cpu().numpy(); v = v.cpu().numpy()
print("oracle", rv.shape, rf.shape, "gpu", v.shape, f.shape)
cells = cell_tables(vol)
cum = 0
n = min(len(f), len(rf))
neq = np.nonzero((f[:n] != rf[:n]).any(1))[0]
first = neq[0] if len(neq) else n
print("first differing face", first)
for (zyx, nt, t, r, s, vals) in cells:
    if cum <= first < cum + max(nt, 1) and nt > 0 or (nt > 0 and cum + nt > first >= cum):
        print("cell", zyx, "oracle nt", nt, "table", t, "row", r, "sub", s, "vals", vals)
        print("oracle faces", rf[cum:cum + nt].tolist())
        print("gpu faces   ", f[cum:cum + nt + 2].tolist())
        break
    cum += nt
