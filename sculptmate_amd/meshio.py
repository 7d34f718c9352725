"""Mesh hand-off formats: Wavefront OBJ with optional per-vertex colours.

Host-side writer for the arrays TSR.run() returns (vertices f32[Nv,3], faces i64[Nf,3], colours f32[Nv,3]|None).
"""
import numpy as np


def write_obj(path, vertices, faces, vertex_colors=None):
    """`v x y z [r g b]` / `f a b c` (1-based), one numpy formatting pass per block."""
    v = np.asarray(vertices, np.float64)
    f = np.asarray(faces, np.int64) + 1
    if vertex_colors is not None:
        v = np.concatenate([v, np.asarray(vertex_colors, np.float64)], 1)
        fmt = "v %.7g %.7g %.7g %.5f %.5f %.5f"
    else:
        fmt = "v %.7g %.7g %.7g"
    with open(path, "w") as fh:
        fh.write("# sculptmate_amd\n")
        np.savetxt(fh, v, fmt=fmt)
        np.savetxt(fh, f, fmt="f %d %d %d")


def read_obj(path):
    vs, cs, fs = [], [], []
    with open(path) as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                vs.append([float(x) for x in p[1:4]])
                if len(p) >= 7:
                    cs.append([float(x) for x in p[4:7]])
            elif p[0] == "f":
                fs.append([int(x.split("/")[0]) - 1 for x in p[1:4]])
    return (np.array(vs, np.float32), np.array(fs, np.int64), np.array(cs, np.float32) if cs else None)
