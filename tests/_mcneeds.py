"""Which lattice values does (Lewiner) marching cubes read?  Test infrastructure for the two-pass density grid.

From the case table of the algorithm (oracle/mc_luts.h: mc_cases, read from scikit-image) and the structure of
oracle/mc_lewiner.c / csrc/mc.hip::classify:
  * cases 1, 2, 5, 8, 9, 11, 14 choose their tiling from the corner signs and place vertices on sign-changing edges only:
    the values read are the END POINTS of the sign-changing lattice edges;
  * cases 3, 4, 6, 7, 10, 12, 13 run face / interior tests on the corner values (and may add the centre vertex): ALL 8 corners.
"""
import os
import re

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# corner k of a cell (csrc/mc.hip::load_cell) -> (d axis0, d axis1, d axis2)
CORNERS = [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)]
AMBIGUOUS_CASES = (3, 4, 6, 7, 10, 12, 13)


def ambiguous_patterns(luts=os.path.join(ROOT, "oracle", "mc_luts.h")):
    """bool[256]: sign patterns (bit k = corner k above the level) whose base case runs tests on the corner values."""
    src = open(luts).read()
    m = re.search(r"static const signed char mc_cases\[512\] = \{(.*?)\};", src, re.S)
    vals = [int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
    assert len(vals) == 512
    return np.array([vals[2 * i] in AMBIGUOUS_CASES for i in range(256)])


def needed_points(above, luts=None):
    """above: bool tensor [n0, n1, n2] (value > level) -> (needed bool [n0, n1, n2], n_active_cells): the lattice points whose
    VALUE marching cubes reads -- end points of sign-changing lattice edges and all corners of cells with an ambiguous pattern."""
    n0, n1, n2 = above.shape
    dev = above.device
    amb_lut = torch.from_numpy(ambiguous_patterns(*( [luts] if luts else [] ))).to(dev)
    sl = lambda d, n: slice(d, n - 1 + d)  # noqa: E731
    idx = torch.zeros((n0 - 1, n1 - 1, n2 - 1), dtype=torch.int64, device=dev)
    for k, (d0, d1, d2) in enumerate(CORNERS):
        idx |= above[sl(d0, n0), sl(d1, n1), sl(d2, n2)].long() << k
    active = (idx != 0) & (idx != 255)
    amb = amb_lut[idx]
    need = torch.zeros((n0, n1, n2), dtype=torch.bool, device=dev)
    for d0, d1, d2 in CORNERS:
        need[sl(d0, n0), sl(d1, n1), sl(d2, n2)] |= amb
    e = above[1:] != above[:-1]
    need[1:] |= e
    need[:-1] |= e
    e = above[:, 1:] != above[:, :-1]
    need[:, 1:] |= e
    need[:, :-1] |= e
    e = above[:, :, 1:] != above[:, :, :-1]
    need[:, :, 1:] |= e
    need[:, :, :-1] |= e
    return need, int(active.sum())
