"""A recording stand-in for Blender's `bpy` module (TEST INFRASTRUCTURE; no arithmetic of the hot path).

It models exactly the part of the API the reference's two mesh sinks touch
  /root/reference/TripoSR/tsr/system.py:127-168        TSR.import_obj_blender
  /root/reference/StableFast/sf3d/system.py:530-598    SF3D.import_mesh_blender
and the vectorised calls the replacements use instead of the per-loop Python loops (`foreach_get` / `foreach_set`),
and records what ends up in the scene: meshes (vertices, polygons, loops), per-loop colours and UVs, objects,
materials with their node graph (node types, links, socket values) and images (pixels, colour space).
`summary()` turns that into plain arrays / JSON-able structures so that two runs can be compared.

tests/golden/make_blender_goldens.py runs the REFERENCE functions against it (build container only) and stores the
summary in tests/golden/blender_sink.npz; tests/test_blender_sinks.py runs sculptmate_amd's sinks against it and
compares.
"""
import sys
import types

import numpy as np


class _Named(dict):
    """bpy collection addressed by name or index, iterable in creation order."""

    def __init__(self):
        super().__init__()
        self._order = []

    def _add(self, name, item):
        self[name] = item
        self._order.append(item)
        return item

    def __getitem__(self, k):
        if isinstance(k, (int, np.integer)):
            return self._order[k]
        return dict.__getitem__(self, k)

    def __iter__(self):
        return iter(list(self._order))

    def __len__(self):
        return len(self._order)


class _Elem:
    """One loop-colour / loop-uv / loop element."""

    def __init__(self, store, i, width=None):
        self._s, self._i = store, i

    @property
    def color(self):
        return self._s.arr[self._i]

    @color.setter
    def color(self, v):
        self._s.arr[self._i] = np.asarray(v, np.float64)

    @property
    def uv(self):
        return self._s.arr[self._i]

    @uv.setter
    def uv(self, v):
        self._s.arr[self._i] = np.asarray(v, np.float64)

    @property
    def vertex_index(self):
        return int(self._s.arr[self._i])


class _ElemArray:
    def __init__(self, n, width, attr):
        self.arr = np.zeros((n, width) if width > 1 else (n,), np.float64 if width > 1 else np.int64)
        self.attr = attr

    def __len__(self):
        return len(self.arr)

    def __getitem__(self, i):
        return _Elem(self, i)

    def __iter__(self):
        return (_Elem(self, i) for i in range(len(self.arr)))

    def foreach_set(self, attr, flat):
        assert attr == self.attr, (attr, self.attr)
        flat = np.asarray(flat)
        assert flat.size == self.arr.size, "foreach_set: %d values for %d slots" % (flat.size, self.arr.size)
        self.arr[...] = flat.reshape(self.arr.shape)

    def foreach_get(self, attr, out):
        assert attr == self.attr, (attr, self.attr)
        out[...] = self.arr.reshape(out.shape)


class _Layer:
    def __init__(self, name, n, width, attr):
        self.name = name
        self.data = _ElemArray(n, width, attr)


class _Layers(_Named):
    def __init__(self, mesh, width, attr):
        super().__init__()
        self._mesh, self._width, self._attr = mesh, width, attr
        self.active = None

    def new(self, name="Layer"):
        layer = self._add(name, _Layer(name, len(self._mesh.loops), self._width, self._attr))
        self.active = layer
        return layer


class _Polygon:
    def __init__(self, start, n):
        self.loop_start, self.loop_total = start, n
        self.loop_indices = range(start, start + n)


class _Mesh:
    def __init__(self, name):
        self.name = name
        self.vertices_co = np.zeros((0, 3))
        self.polygons = []
        self.loops = _ElemArray(0, 1, "vertex_index")
        self.vertex_colors = _Layers(self, 4, "color")
        self.uv_layers = _Layers(self, 2, "uv")
        self.materials = []

    def from_pydata(self, verts, edges, faces):
        assert len(edges) == 0
        self.vertices_co = np.asarray(verts, np.float64).reshape(-1, 3)
        faces = [list(map(int, f)) for f in faces]
        self.polygons, idx, start = [], [], 0
        for f in faces:
            self.polygons.append(_Polygon(start, len(f)))
            idx.extend(f)
            start += len(f)
        self.loops = _ElemArray(len(idx), 1, "vertex_index")
        self.loops.arr[...] = np.asarray(idx, np.int64)
        assert self.loops.arr.size == 0 or (self.loops.arr.min() >= 0 and self.loops.arr.max() < len(self.vertices_co))


class _Object:
    def __init__(self, name, object_data):
        self.name, self.data = name, object_data
        self.selected = False

    def select_set(self, state):
        self.selected = bool(state)


class _Socket:
    def __init__(self, node, name):
        self.node, self.name = node, name
        self.default_value = None


class _Sockets(dict):
    def __init__(self, node):
        super().__init__()
        self._node = node

    def __missing__(self, k):
        s = self[k] = _Socket(self._node, k)
        return s


class _Node:
    def __init__(self, type_):
        self.type = type_
        self.inputs, self.outputs = _Sockets(self), _Sockets(self)
        self.layer_name = None
        self.image = None
        self.location = None


class _Nodes(list):
    def __iter__(self):
        # Blender's collection iterator survives removal of the element it has just returned (the reference's
        # `for node in nodes: nodes.remove(node)` empties the tree there): iterate over a snapshot
        return iter(list(self[:]))

    def new(self, type=None, *a):  # noqa: A002 (bpy's own keyword)
        if type is None:
            type = a[0]
        n = _Node(type)
        self.append(n)
        return n

    def remove(self, node):
        list.remove(self, node)

    def clear(self):
        del self[:]


class _Links(list):
    def new(self, a, b):
        self.append((a, b))


class _NodeTree:
    def __init__(self):
        self.nodes, self.links = _Nodes(), _Links()
        # what Blender creates for use_nodes = True: the sinks must remove these
        self.nodes.new(type="ShaderNodeBsdfPrincipled")
        self.nodes.new(type="ShaderNodeOutputMaterial")


class _Material:
    def __init__(self, name):
        self.name = name
        self.use_nodes = False
        self.node_tree = _NodeTree()


class _Pixels:
    def __init__(self, n):
        self.arr = np.zeros(n, np.float64)

    def foreach_set(self, flat):
        flat = np.asarray(flat, np.float64).ravel()
        assert flat.size == self.arr.size
        self.arr[...] = flat


class _Image:
    def __init__(self, name, width, height):
        self.name, self.size = name, (width, height)
        self._pixels = _Pixels(width * height * 4)
        self.colorspace_settings = types.SimpleNamespace(name="sRGB")

    @property
    def pixels(self):
        return self._pixels

    @pixels.setter
    def pixels(self, v):
        v = np.asarray(v, np.float64).ravel()
        assert v.size == self._pixels.arr.size, "pixels: %d values for a %dx%d RGBA image" % ((v.size,) + self.size)
        self._pixels.arr[...] = v


class _Factory(_Named):
    def __init__(self, ctor):
        super().__init__()
        self._ctor = ctor

    def new(self, *a, **k):
        name = k.get("name", a[0] if a else "item")
        args = [x for x in a[1:]] if a else []
        kw = {kk: vv for kk, vv in k.items() if kk != "name"}
        return self._add(name if name not in self else "%s.%03d" % (name, len(self)), self._ctor(name, *args, **kw))


def make():
    """A fresh fake `bpy` module."""
    bpy = types.ModuleType("bpy")
    bpy.data = types.SimpleNamespace(meshes=_Factory(_Mesh), objects=_Factory(_Object), materials=_Factory(_Material),
                                     images=_Factory(_Image))
    linked = []
    objects = types.SimpleNamespace(link=linked.append, linked=linked)
    bpy.context = types.SimpleNamespace(collection=types.SimpleNamespace(objects=objects),
                                        view_layer=types.SimpleNamespace(objects=types.SimpleNamespace(active=None)))
    return bpy


def install(bpy=None):
    bpy = bpy or make()
    sys.modules["bpy"] = bpy
    return bpy


def summary(bpy):
    """Everything the sinks left in the scene, as {key: ndarray | str(JSON)}."""
    import json

    out, meta = {}, {"meshes": [], "objects": [], "materials": [], "images": [], "linked": [], "active": None}
    for mi, m in enumerate(bpy.data.meshes):
        k = "mesh%d." % mi
        out[k + "vertices"] = m.vertices_co.astype(np.float32)
        out[k + "loop_vertex_index"] = m.loops.arr.astype(np.int64)
        out[k + "poly_loop_start"] = np.array([p.loop_start for p in m.polygons], np.int64)
        out[k + "poly_loop_total"] = np.array([p.loop_total for p in m.polygons], np.int64)
        layers = {"vertex_colors": [], "uv_layers": []}
        for kind in layers:
            for layer in getattr(m, kind):
                layers[kind].append(layer.name)
                out[k + "%s.%s" % (kind, layer.name)] = layer.data.arr.astype(np.float32)
        meta["meshes"].append({"name": m.name, "materials": [x.name for x in m.materials], **layers,
                               "active_uv": None if m.uv_layers.active is None else m.uv_layers.active.name})
    for o in bpy.data.objects:
        meta["objects"].append({"name": o.name, "data": o.data.name, "selected": o.selected})
    meta["linked"] = [o.name for o in bpy.context.collection.objects.linked]
    act = bpy.context.view_layer.objects.active
    meta["active"] = None if act is None else act.name
    for mat in bpy.data.materials:
        nodes = list(mat.node_tree.nodes)
        ids, seen = {}, {}
        for n in nodes:  # node id = type + occurrence, independent of creation order across types
            seen[n.type] = seen.get(n.type, 0) + 1
            ids[id(n)] = "%s#%d" % (n.type, seen[n.type] - 1)
        nd = []
        for n in nodes:
            vals = {s.name: float(s.default_value) for s in n.inputs.values() if s.default_value is not None}
            nd.append({"id": ids[id(n)], "layer_name": n.layer_name, "image": None if n.image is None else n.image.name,
                       "input_values": vals})
        nd.sort(key=lambda d: d["id"])
        links = sorted([ids[id(a.node)], a.name, ids[id(b.node)], b.name] for a, b in mat.node_tree.links)
        meta["materials"].append({"name": mat.name, "use_nodes": bool(mat.use_nodes), "nodes": nd, "links": links})
    for ii, im in enumerate(bpy.data.images):
        out["image%d.pixels" % ii] = im.pixels.arr.astype(np.float32)
        meta["images"].append({"name": im.name, "size": list(im.size), "colorspace": im.colorspace_settings.name})
    out["meta"] = np.array(json.dumps(meta, sort_keys=True))
    return out
