"""Mesh hand-off formats: Wavefront OBJ, binary PLY and binary glTF (.glb).

Host-side writers for the arrays TSR.run() returns (vertices f32[Nv,3], faces i64[Nf,3], colours f32[Nv,3]|None)
and for the dict SF3D.run_image() returns (vertices, faces, uvs, basecolor_tex, bump_tex, roughness, metallic):
the reference hands these to Blender only (tsr/system.py:127-169, sf3d/system.py:504-560); upstream TripoSR / SF3D
users export them through trimesh, which is not a dependency here.
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


# ---------------------------------------------------------------------------------------------------------
# Binary glTF 2.0 (.glb): the container upstream users get from trimesh's `export("mesh.glb")`.  One buffer, one
# mesh primitive; float32 POSITION / NORMAL / TEXCOORD_0 / COLOR_0, uint32 indices, optional PBR textures
# (base colour, normal map) embedded as PNG.  No third-party dependency: the PNG encoder is zlib + CRC.
# ---------------------------------------------------------------------------------------------------------
import json
import struct
import zlib

_GLB_MAGIC, _CHUNK_JSON, _CHUNK_BIN = 0x46546C67, 0x4E4F534A, 0x004E4942
_FLOAT, _UINT32 = 5126, 5125
_ARRAY_BUFFER, _ELEMENT_ARRAY_BUFFER = 34962, 34963


def encode_png(image):
    """uint8 [H,W,C] with C in {1,3,4} (or float in [0,1]) -> PNG bytes (filter 0 on every scanline)."""
    a = np.asarray(image)
    if a.dtype != np.uint8:
        a = np.round(np.clip(a.astype(np.float64), 0.0, 1.0) * 255.0).astype(np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    colour_type = {1: 0, 3: 2, 4: 6}[c]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * c)], 1).tobytes()

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, colour_type, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def decode_png(data):
    """Inverse of encode_png for the files it writes (8-bit, non-interlaced, filter 0)."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, head = 8, b"", None
    while pos < len(data):
        (n,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            head = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, colour_type = head[:4]
    c = {0: 1, 2: 3, 6: 4}[colour_type]
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * c)
    if depth != 8 or rows[:, 0].any():
        raise ValueError("decode_png reads only what encode_png writes")
    return rows[:, 1:].reshape(h, w, c)


def write_glb(path, vertices, faces, vertex_colors=None, normals=None, uvs=None, basecolor_tex=None,
              normal_tex=None, roughness=None, metallic=None, name="mesh", uv_origin="bottom_left"):
    """Write one triangle mesh as binary glTF.  `uvs` as SF3D.run_image returns them have their origin at the BOTTOM-left
    of the top-down texture image (the baker samples row y at v = 1 - y/R, texture_baker/common.py:104-142, and the
    reference hands them to Blender unchanged, sf3d/system.py:539-545); glTF puts the origin at the top-left, so
    TEXCOORD_0 is written as (u, 1 - v) unless uv_origin="top_left".  roughness / metallic are SF3D's scalar factors."""
    if uv_origin not in ("bottom_left", "top_left"):
        raise ValueError("uv_origin must be 'bottom_left' or 'top_left'")
    v = np.ascontiguousarray(vertices, np.float32)
    idx = np.ascontiguousarray(faces, np.int64)
    if idx.size and (idx.min() < 0 or idx.max() >= len(v)):
        raise ValueError("write_glb: face index out of range")
    blob, views, accessors, images = bytearray(), [], [], []

    def add_view(raw, target=None):
        while len(blob) % 4:
            blob.append(0)
        view = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(raw)}
        if target is not None:
            view["target"] = target
        blob.extend(raw)
        views.append(view)
        return len(views) - 1

    def add_accessor(arr, kind, ctype, target, bounds=False):
        acc = {"bufferView": add_view(arr.tobytes(), target), "componentType": ctype, "count": int(arr.shape[0]),
               "type": kind}
        if bounds and arr.shape[0]:
            acc["min"], acc["max"] = [float(x) for x in arr.min(0)], [float(x) for x in arr.max(0)]
        accessors.append(acc)
        return len(accessors) - 1

    attributes = {"POSITION": add_accessor(v, "VEC3", _FLOAT, _ARRAY_BUFFER, bounds=True)}
    if normals is not None:
        attributes["NORMAL"] = add_accessor(np.ascontiguousarray(normals, np.float32), "VEC3", _FLOAT, _ARRAY_BUFFER)
    if uvs is not None:
        st = np.array(uvs, np.float32)
        if uv_origin == "bottom_left":
            st[:, 1] = np.float32(1.0) - st[:, 1]
        attributes["TEXCOORD_0"] = add_accessor(np.ascontiguousarray(st), "VEC2", _FLOAT, _ARRAY_BUFFER)
    if vertex_colors is not None:
        col = np.ascontiguousarray(vertex_colors, np.float32)
        attributes["COLOR_0"] = add_accessor(col, "VEC3" if col.shape[1] == 3 else "VEC4", _FLOAT, _ARRAY_BUFFER)
    indices = add_accessor(idx.astype(np.uint32).reshape(-1), "SCALAR", _UINT32, _ELEMENT_ARRAY_BUFFER)

    pbr = {}
    material = {"name": name + "_material", "pbrMetallicRoughness": pbr, "doubleSided": True}
    textures = []

    def add_texture(img):
        images.append({"bufferView": add_view(encode_png(img)), "mimeType": "image/png"})
        textures.append({"source": len(images) - 1, "sampler": 0})
        return {"index": len(textures) - 1}

    if basecolor_tex is not None:
        pbr["baseColorTexture"] = add_texture(basecolor_tex)
    if normal_tex is not None:
        material["normalTexture"] = add_texture(normal_tex)
    pbr["roughnessFactor"] = 1.0 if roughness is None else float(np.asarray(roughness).reshape(-1)[0])
    pbr["metallicFactor"] = 0.0 if metallic is None else float(np.asarray(metallic).reshape(-1)[0])

    doc = {
        "asset": {"version": "2.0", "generator": "sculptmate_amd"},
        "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0, "name": name}],
        "meshes": [{"name": name, "primitives": [{"attributes": attributes, "indices": indices, "material": 0,
                                                   "mode": 4}]}],
        "materials": [material], "accessors": accessors, "bufferViews": views,
    }
    if textures:
        doc["textures"], doc["images"] = textures, images
        doc["samplers"] = [{"magFilter": 9729, "minFilter": 9987, "wrapS": 33071, "wrapT": 33071}]
    while len(blob) % 4:
        blob.append(0)
    doc["buffers"] = [{"byteLength": len(blob)}]
    text = json.dumps(doc, separators=(",", ":")).encode()
    text += b" " * (-len(text) % 4)
    total = 12 + 8 + len(text) + 8 + len(blob)
    with open(path, "wb") as fh:
        fh.write(struct.pack("<III", _GLB_MAGIC, 2, total))
        fh.write(struct.pack("<II", len(text), _CHUNK_JSON) + text)
        fh.write(struct.pack("<II", len(blob), _CHUNK_BIN) + bytes(blob))


def read_glb(path, uv_origin="bottom_left"):
    """Read back what write_glb wrote -> dict(vertices, faces, normals, uvs, vertex_colors, basecolor_tex, normal_tex,
    roughness, metallic); uvs are returned in `uv_origin` convention (see write_glb).  Not a general glTF loader: one
    buffer, one primitive, tightly packed accessors."""
    with open(path, "rb") as fh:
        data = fh.read()
    magic, version, total = struct.unpack("<III", data[:12])
    if magic != _GLB_MAGIC or version != 2 or total != len(data):
        raise ValueError("not a glTF 2.0 binary container")
    n_json, tag = struct.unpack("<II", data[12:20])
    assert tag == _CHUNK_JSON
    doc = json.loads(data[20:20 + n_json])
    n_bin, tag = struct.unpack("<II", data[20 + n_json:28 + n_json])
    assert tag == _CHUNK_BIN
    blob = data[28 + n_json:28 + n_json + n_bin]

    def view_bytes(i):
        bv = doc["bufferViews"][i]
        return blob[bv["byteOffset"]:bv["byteOffset"] + bv["byteLength"]]

    def accessor(i):
        acc = doc["accessors"][i]
        width = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}[acc["type"]]
        dt = {_FLOAT: np.float32, _UINT32: np.uint32}[acc["componentType"]]
        return np.frombuffer(view_bytes(acc["bufferView"]), dt, acc["count"] * width).reshape(acc["count"], width)

    prim = doc["meshes"][0]["primitives"][0]
    att = prim["attributes"]
    mat = doc["materials"][prim["material"]]
    pbr = mat["pbrMetallicRoughness"]

    def texture(ref):
        if ref is None:
            return None
        return decode_png(view_bytes(doc["images"][doc["textures"][ref["index"]]["source"]]["bufferView"]))

    st = np.array(accessor(att["TEXCOORD_0"])) if "TEXCOORD_0" in att else None
    if st is not None and uv_origin == "bottom_left":
        st[:, 1] = np.float32(1.0) - st[:, 1]
    return {
        "vertices": accessor(att["POSITION"]),
        "faces": accessor(prim["indices"]).reshape(-1, 3).astype(np.int64),
        "normals": accessor(att["NORMAL"]) if "NORMAL" in att else None,
        "uvs": st,
        "vertex_colors": accessor(att["COLOR_0"]) if "COLOR_0" in att else None,
        "basecolor_tex": texture(pbr.get("baseColorTexture")),
        "normal_tex": texture(mat.get("normalTexture")),
        "roughness": pbr["roughnessFactor"], "metallic": pbr["metallicFactor"],
    }


def _ply_layout(vertices, faces, vertex_colors):
    """(header bytes, vertex block as a C-contiguous array whose bytes ARE the block, int faces [Nf,3]) of write_ply's file."""
    v = np.ascontiguousarray(vertices, np.float32)
    f = np.asarray(faces)
    if f.dtype.kind not in "iu":
        f = f.astype(np.int64)
    head = ["ply", "format binary_little_endian 1.0", "comment sculptmate_amd", "element vertex %d" % len(v),
            "property float x", "property float y", "property float z"]
    if vertex_colors is not None:
        head += ["property uchar red", "property uchar green", "property uchar blue"]
        c8 = np.round(np.clip(np.asarray(vertex_colors, np.float64), 0, 1) * 255).astype(np.uint8)
        vblock = np.empty((len(v), 15), np.uint8)          # float32 xyz + uchar rgb, packed
        vblock[:, :12] = v.view(np.uint8).reshape(len(v), 12)
        vblock[:, 12:] = c8
    else:
        vblock = v                                           # the raw [Nv,3] float32 buffer is the vertex block
    head += ["element face %d" % len(f), "property list uchar int vertex_indices", "end_header"]
    return ("\n".join(head) + "\n").encode(), vblock, f


_tls = None


def _ply_face_records(f):
    """int [n,3] -> uint8 [n,13]: `uchar 3` + three little-endian int32 per face.  int64 faces (what TSR.run returns) go
    through the library's host-side packer (sculpt_ply_face_records: one pass, and ctypes drops the GIL around it, so chunks
    run in parallel); anything else through NumPy.  The result lives in a per-thread scratch buffer that the next call on the
    same thread overwrites (fresh pages per chunk cost more than the packing: the page faults of all threads serialise)."""
    global _tls
    if _tls is None:
        import threading

        _tls = threading.local()
    buf = getattr(_tls, "buf", None)
    if buf is None or len(buf) < len(f):
        buf = _tls.buf = np.empty((max(len(f), PLY_CHUNK_FACES), 13), np.uint8)
    rec = buf[:len(f)]
    if f.dtype == np.int64 and f.flags.c_contiguous and len(f):
        from . import _lib

        _lib.check(_lib.lib.sculpt_ply_face_records(f.ctypes.data, len(f), rec.ctypes.data))
        return rec
    rec[:, 0] = 3
    rec[:, 1:] = np.ascontiguousarray(f, "<i4").view(np.uint8).reshape(len(f), 12)
    return rec


PLY_CHUNK_FACES = 1 << 18   # faces converted + written per task (3.4 MB of records)


def write_ply(path, vertices, faces, vertex_colors=None, pool=None):
    """Binary little-endian PLY: float32 xyz, optional uchar rgb, faces as `uchar 3` + int32 triplets.
    The file is laid out up front (header, vertex block, 13-byte face records) and written with positioned writes, the face
    records converted chunk by chunk; with `pool` (a concurrent.futures executor) the chunks are converted and written in
    parallel -- NumPy copies and os.pwrite release the GIL -- and the same bytes land in the file either way."""
    import os

    head, vblock, f = _ply_layout(vertices, faces, vertex_colors)
    v_off = len(head)
    f_off = v_off + vblock.nbytes
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        os.ftruncate(fd, f_off + 13 * len(f))

        def put(data, off):
            mv = memoryview(data).cast("B")
            while len(mv):
                n = os.pwrite(fd, mv, off)
                mv, off = mv[n:], off + n

        def face_chunk(a):
            put(_ply_face_records(f[a:a + PLY_CHUNK_FACES]), f_off + 13 * a)

        tasks = [lambda: put(head, 0)]
        step = 4 << 20                                       # vertex block in 4 MB pieces
        vb = memoryview(vblock).cast("B")
        tasks += [lambda a=a: put(vb[a:a + step], v_off + a) for a in range(0, len(vb), step)]
        tasks += [lambda a=a: face_chunk(a) for a in range(0, len(f), PLY_CHUNK_FACES)]
        if pool is None:
            for t in tasks:
                t()
        else:
            for fut in [pool.submit(t) for t in tasks]:
                fut.result()
    finally:
        os.close(fd)


def write_npz(path, vertices, faces, vertex_colors=None):
    """Raw arrays (np.savez, uncompressed): vertices f32, faces as given (int64 from TSR.run), vertex_colors if any."""
    arrays = {"vertices": np.asarray(vertices, np.float32), "faces": np.asarray(faces)}
    if vertex_colors is not None:
        arrays["vertex_colors"] = np.asarray(vertex_colors, np.float32)
    with open(path, "wb") as fh:
        np.savez(fh, **arrays)


def read_ply(path):
    """Read back what write_ply wrote -> (vertices f32, faces i64, colours f32 in [0,1] | None)."""
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    head = data[:end].decode().splitlines()
    nv = int([h for h in head if h.startswith("element vertex")][0].split()[-1])
    nf = int([h for h in head if h.startswith("element face")][0].split()[-1])
    coloured = any(h == "property uchar red" for h in head)
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")] + ([("red", "u1"), ("green", "u1"), ("blue", "u1")] if coloured else [])
    vdt = np.dtype(fields)
    vrec = np.frombuffer(data, vdt, nv, end)
    frec = np.frombuffer(data, np.dtype([("n", "u1"), ("i", "<i4", (3,))]), nf, end + nv * vdt.itemsize)
    v = np.stack([vrec["x"], vrec["y"], vrec["z"]], 1)
    c = np.stack([vrec["red"], vrec["green"], vrec["blue"]], 1).astype(np.float32) / 255.0 if coloured else None
    return v, frec["i"].astype(np.int64), c


def write_sf3d_glb(path, mesh, name="mesh"):
    """The dict SF3D.run_image returns (vertices, faces, uvs, basecolor_tex, bump_tex, roughness, metallic) -> .glb."""
    def image(x):
        return None if x is None else np.asarray(x)[..., :3]

    write_glb(path, mesh["vertices"], mesh["faces"], uvs=mesh.get("uvs"), basecolor_tex=image(mesh.get("basecolor_tex")),
              normal_tex=image(mesh.get("bump_tex")), roughness=mesh.get("roughness"), metallic=mesh.get("metallic"),
              name=name)
