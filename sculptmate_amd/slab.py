"""BASELINE config 5: one image at 512^3, the voxel grid split into slabs along the slowest lattice axis
(reference axis 0 = "x", isosurface.py:34-37) over the ranks, RCCL all-gather of the per-slab triangles.

Partition: rank r owns cell layers [c0, c1) and evaluates lattice planes [c0, c1]
(the extra plane is a recomputed halo -- density is a pure function of position, nothing is exchanged
before marching cubes).  Slabs are contiguous in memory and in scikit-image's sweep order, so concatenating
the per-slab meshes in rank order reproduces the single-GPU mesh exactly (vertices, faces AND order):
  * a vertex on an x/y edge of a boundary plane is created by the LOWER slab (first cell to touch it);
    the upper slab does not emit it and its faces carry the reference -(1 + slot), resolved after the
    gather through the lower slab's top_plane_map;
  * everything else is local; ids only need the rank's vertex base offset.
The only exchange step is the gather of (verts, faces, top_plane_map) -- one padded all_gather each.
"""
import torch

from . import ops


def slab_ranges(R, world):
    """[(c0, c1)] cell-layer ranges per rank, as equal as possible; planes evaluated = [c0, c1]."""
    layers = R - 1
    base, rem = divmod(layers, world)
    out, c = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((c, c + n))
        c += n
    return out


def extract_slab(planes, mlp, R, rank, world, radius=0.87, density_bias=-1.0, threshold=25.0, precision="bf16l3", run=None):
    """Density + marching cubes of this rank's slab -> dict of device tensors (local ids, refs < 0).
    run(x_begin, x_end, mc) (optional): evaluates lattice planes [x_begin, x_end) and returns mc(volume) -- TSR passes its two-pass
    grid with the run-time guard here (TSR._extract_filtered); default: the full evaluation in `precision`."""
    c0, c1 = slab_ranges(R, world)[rank]
    if c1 <= c0:
        dev = planes.device
        return dict(verts=torch.empty((0, 3), device=dev), faces=torch.empty((0, 3), dtype=torch.int64, device=dev),
                    top=torch.full((2, R, R), -1, dtype=torch.int32, device=dev), minmax=(float("inf"), float("-inf")))

    def mc(vol):
        return ops.marching_cubes(vol.view(c1 - c0 + 1, R, R), 0.0, reference_order=True, vert_div=R - 1.0,
                                  vert_mul=radius - (-radius), vert_add=-radius,
                                  slab=dict(axis0_offset=c0, halo_low=rank > 0 and c0 > 0))

    if run is not None:
        v, f, top, mm = run(c0, c1 + 1, mc)
    else:
        v, f, top, mm = mc(ops.density_grid(planes, mlp, R, radius=radius, density_bias=density_bias, x_begin=c0, x_end=c1 + 1,
                                            out_add=-threshold, precision=precision))
    return dict(verts=v, faces=f, top=top, minmax=mm)


def assemble(parts):
    """parts: per-rank dicts (verts [nv,3], faces [nf,3] int64 with refs, top [2,n1,n2]) in rank order, all on
    one device -> (verts, faces) identical to the single-volume marching cubes."""
    verts, faces = [], []
    base = 0
    prev_top, prev_base = None, 0
    for p in parts:
        f = p["faces"].clone()
        neg = f < 0
        if neg.any():
            if prev_top is None:
                raise RuntimeError("slab assemble: reference into a missing previous slab")
            slot = (-f[neg] - 1).long()
            ref = prev_top.reshape(-1)[slot].long()
            if (ref < 0).any():
                raise RuntimeError("slab assemble: unresolved boundary vertex")
            resolved = ref + prev_base
        f = torch.where(neg, torch.zeros_like(f), f + base)
        if neg.any():
            f[neg] = resolved
        faces.append(f)
        verts.append(p["verts"])
        if p["verts"].shape[0] or p["top"] is not None:
            prev_top, prev_base = p["top"], base
        base += p["verts"].shape[0]
    return torch.cat(verts, 0), torch.cat(faces, 0)


def _gather_padded(t, device):
    """all_gather of a variable-length tensor: sizes first, then one padded gather."""
    import torch.distributed as dist

    world = dist.get_world_size()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=device)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [o[:s] for o, s in zip(out, sizes)]


def gather_and_assemble(part, device):
    """Every rank contributes its slab; every rank returns the full mesh (the exchange step)."""
    vs = _gather_padded(part["verts"], device)
    fs = _gather_padded(part["faces"], device)
    import torch.distributed as dist

    tops = [torch.empty_like(part["top"]) for _ in range(dist.get_world_size())]
    dist.all_gather(tops, part["top"].contiguous())
    mm = torch.tensor([part["minmax"][0], -part["minmax"][1]], dtype=torch.float32, device=device)
    dist.all_reduce(mm, op=dist.ReduceOp.MIN)
    v, f = assemble([dict(verts=a, faces=b, top=c) for a, b, c in zip(vs, fs, tops)])
    _check_range(float(mm[0]), -float(mm[1]), v.shape[0])
    return v, f


def _check_range(mn, mx, nv, level_offset=0.0):
    # skimage semantics on the whole volume (isosurface.py:46-48 -> marching_cubes(level, 0.0))
    if 0.0 < mn or 0.0 > mx:
        raise ValueError("Surface level must be within volume data range.")
    if nv == 0:
        raise RuntimeError("No surface found at the given iso value.")


def extract_mesh_slabs_local(planes, mlp, R, world, **kw):
    """All slabs evaluated one after the other on THIS GPU and assembled (what N ranks would produce):
    used by the parity tests and as the single-GPU path for grids too large for one pass."""
    parts = [extract_slab(planes, mlp, R, r, world, **kw) for r in range(world)]
    v, f = assemble(parts)
    _check_range(min(p["minmax"][0] for p in parts), max(p["minmax"][1] for p in parts), v.shape[0])
    return v, f
