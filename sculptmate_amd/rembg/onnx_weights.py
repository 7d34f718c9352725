"""Read U^2-Net's parameters out of `u2net.onnx` -- the file the reference hands to onnxruntime
(rembg/sessions/base.py:38-42) -- without onnx / onnxruntime / protobuf schemas: an ONNX file is a protobuf message
and only its wire format is needed to reach the graph's nodes and initialisers.

Field numbers (onnx.proto, the public schema): ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5;
NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4; TensorProto.dims = 1, .data_type = 2, .float_data = 4,
.name = 8, .raw_data = 9.  Only FLOAT (1) tensors stored inline are accepted.

Two exporter habits are handled when mapping initialisers to the authors' parameter names (spec.param_spec()):
  * initialisers keep the PyTorch names ("stage1.rebnconvin.conv_s1.weight", "...bn_s1.running_mean"): used directly;
  * initialisers are anonymous (numeric names, BatchNorm possibly folded into the convolution at export): the Conv
    nodes are taken in graph order, which is the network's forward order == spec order (stages, side convolutions,
    fusion convolution), each with the BatchNormalization node that consumes its output, if any.  Every shape is
    checked against the spec, so a graph that is not this U^2-Net is rejected instead of mis-mapped.
"""
import numpy as np

from .spec import BN_EPS, SIDES, STAGES, param_spec, rsu_layers

_FLOAT = 1


class OnnxFormatError(ValueError):
    pass


def _varint(buf, pos):
    out = shift = 0
    while True:
        if pos >= len(buf):
            raise OnnxFormatError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise OnnxFormatError("varint longer than 64 bits")


def fields(buf):
    """Yield (field number, wire type, value) of one protobuf message; length-delimited values are memoryviews."""
    buf = memoryview(buf)
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wire = key >> 3, key & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 1:
            val, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wire == 2:
            n, pos = _varint(buf, pos)
            if pos + n > end:
                raise OnnxFormatError("length-delimited field runs past the end of its message")
            val, pos = buf[pos:pos + n], pos + n
        elif wire == 5:
            val, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise OnnxFormatError("unsupported protobuf wire type %d" % wire)
        yield num, wire, val


def _tensor(buf):
    dims, dtype, name, raw, floats = [], None, "", None, []
    for num, wire, val in fields(buf):
        if num == 1:
            if wire == 2:  # packed
                p = 0
                while p < len(val):
                    d, p = _varint(val, p)
                    dims.append(d)
            else:
                dims.append(val)
        elif num == 2:
            dtype = val
        elif num == 4:
            if wire == 2:
                floats.append(np.frombuffer(val, "<f4"))
            else:
                floats.append(np.frombuffer(val, "<f4", 1))
        elif num == 8:
            name = bytes(val).decode()
        elif num == 9:
            raw = val
        elif num == 13 or num == 14:
            raise OnnxFormatError("initializer %r keeps its data in an external file; only inline data is read" % name)
    if dtype != _FLOAT:
        return name, None  # int64 shape constants etc.: not parameters
    data = np.frombuffer(raw, "<f4") if raw is not None else (np.concatenate(floats) if floats else np.zeros(0, "<f4"))
    count = int(np.prod(dims)) if dims else 1
    if data.size != count:
        raise OnnxFormatError("initializer %r: %d values for dims %s" % (name, data.size, dims))
    return name, data.reshape(dims).astype(np.float32)


def _node(buf):
    ins, outs, op, name = [], [], "", ""
    for num, _wire, val in fields(buf):
        if num == 1:
            ins.append(bytes(val).decode())
        elif num == 2:
            outs.append(bytes(val).decode())
        elif num == 3:
            name = bytes(val).decode()
        elif num == 4:
            op = bytes(val).decode()
    return {"op": op, "name": name, "inputs": ins, "outputs": outs}


def read_graph(path_or_bytes):
    """-> (nodes in file order [dict(op, name, inputs, outputs)], {initializer name: float32 array})."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        data = path_or_bytes
    else:
        with open(path_or_bytes, "rb") as fh:
            data = fh.read()
    graph = None
    for num, wire, val in fields(data):
        if num == 7 and wire == 2:
            graph = val
    if graph is None:
        raise OnnxFormatError("no GraphProto in the file (not an ONNX model?)")
    nodes, inits = [], {}
    for num, wire, val in fields(graph):
        if num == 1 and wire == 2:
            nodes.append(_node(val))
        elif num == 5 and wire == 2:
            name, arr = _tensor(val)
            if arr is not None:
                inits[name] = arr
    return nodes, inits


def u2net_state_dict(path_or_bytes):
    """`u2net.onnx` -> {authors' parameter name: float32 array}, loadable by U2Net.load_state_dict."""
    spec = param_spec()
    nodes, inits = read_graph(path_or_bytes)
    if all(k in inits for k in spec):
        sd = {k: inits[k] for k in spec}
    else:
        sd = _by_graph_order(nodes, inits, spec)
    for k, shp in spec.items():
        if tuple(sd[k].shape) != tuple(shp):
            raise OnnxFormatError("%s has shape %s in the ONNX graph, U^2-Net needs %s" % (k, tuple(sd[k].shape), shp))
    return sd


def _by_graph_order(nodes, inits, spec):
    consumer = {}
    for n in nodes:
        for name in n["inputs"]:
            consumer.setdefault(name, n)
    convs = [n for n in nodes if n["op"] == "Conv"]
    # forward order of the convolutions: every REBNCONV of every stage, the six side convolutions, the fusion convolution
    prefixes = ["%s.%s.conv_s1." % (stage, layer[0]) for stage, kind, cin, mid, cout in STAGES
                for layer in rsu_layers(kind, cin, mid, cout)]
    prefixes += [side + "." for side, _c in SIDES] + ["outconv."]
    if len(convs) != len(prefixes):
        raise OnnxFormatError("the graph has %d Conv nodes, U^2-Net has %d" % (len(convs), len(prefixes)))
    sd = {}
    for node, p in zip(convs, prefixes):
        params = [inits.get(x) for x in node["inputs"][1:]]
        if not params or params[0] is None:
            raise OnnxFormatError("Conv node %r: weight is not an initializer" % node["name"])
        w = params[0]
        b = params[1] if len(params) > 1 and params[1] is not None else np.zeros(w.shape[0], np.float32)
        sd[p + "weight"], sd[p + "bias"] = w, b
        if not p.endswith("conv_s1."):
            continue
        bn_p = p[:-len("conv_s1.")] + "bn_s1."
        nxt = consumer.get(node["outputs"][0])
        if nxt is not None and nxt["op"] == "BatchNormalization":
            vals = [inits.get(x) for x in nxt["inputs"][1:5]]
            if len(vals) != 4 or any(v is None for v in vals):
                raise OnnxFormatError("BatchNormalization after %r: parameters are not initializers" % node["name"])
            sd[bn_p + "weight"], sd[bn_p + "bias"], sd[bn_p + "running_mean"], sd[bn_p + "running_var"] = vals
        else:  # folded into the convolution by the exporter: an identity BatchNorm keeps the parameter layout
            co = w.shape[0]
            sd[bn_p + "weight"] = np.ones(co, np.float32)
            sd[bn_p + "bias"] = np.zeros(co, np.float32)
            sd[bn_p + "running_mean"] = np.zeros(co, np.float32)
            sd[bn_p + "running_var"] = np.full(co, 1.0 - BN_EPS, np.float32)
    return sd


# ---------------------------------------------------------------------------------------------------------------------
# Writer for the same subset (tests, and converting a state dict for tools that expect the ONNX container's weights).
def _key(num, wire):
    return _enc_varint((num << 3) | wire)


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num, payload):
    return _key(num, 2) + _enc_varint(len(payload)) + bytes(payload)


def encode_tensor(name, arr, raw=True):
    arr = np.ascontiguousarray(arr, "<f4")
    body = b"".join(_key(1, 0) + _enc_varint(int(d)) for d in arr.shape) + _key(2, 0) + _enc_varint(_FLOAT)
    body += _ld(8, name.encode())
    body += _ld(9, arr.tobytes()) if raw else _ld(4, arr.tobytes())
    return body


def encode_node(op, inputs, outputs, name=""):
    body = b"".join(_ld(1, s.encode()) for s in inputs) + b"".join(_ld(2, s.encode()) for s in outputs)
    return body + _ld(3, name.encode()) + _ld(4, op.encode())


def encode_model(nodes, tensors):
    """nodes: [encode_node(...)], tensors: [encode_tensor(...)] -> ModelProto bytes (ir_version 7)."""
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"u2net") + b"".join(_ld(5, t) for t in tensors)
    return _key(1, 0) + _enc_varint(7) + _ld(7, graph)
