"""sculptmate_amd/rembg/onnx_weights.py: U^2-Net parameters out of an ONNX container without onnx/onnxruntime.
The reader is checked against files serialised by the official protobuf runtime (an independent writer) built from
the public onnx.proto field numbers, and against its own writer."""
import numpy as np
import pytest

from sculptmate_amd.rembg import onnx_weights as ow
from sculptmate_amd.rembg.spec import BN_EPS, SIDES, STAGES, param_spec, rsu_layers


def _protobuf_classes():
    """ModelProto / GraphProto / NodeProto / TensorProto subset declared through descriptor_pb2 (field numbers of onnx.proto)."""
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="onnx_subset.proto", package="onnx_subset", syntax="proto2")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, extra in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if "type_name" in extra:
                f.type_name = extra["type_name"]
            if extra.get("packed"):
                f.options.packed = True

    REP, OPT = F.LABEL_REPEATED, F.LABEL_OPTIONAL
    msg("TensorProto", [("dims", 1, F.TYPE_INT64, REP, {}), ("data_type", 2, F.TYPE_INT32, OPT, {}),
                        ("float_data", 4, F.TYPE_FLOAT, REP, {"packed": True}), ("int64_data", 7, F.TYPE_INT64, REP, {"packed": True}),
                        ("name", 8, F.TYPE_STRING, OPT, {}), ("raw_data", 9, F.TYPE_BYTES, OPT, {})])
    msg("NodeProto", [("input", 1, F.TYPE_STRING, REP, {}), ("output", 2, F.TYPE_STRING, REP, {}),
                      ("name", 3, F.TYPE_STRING, OPT, {}), ("op_type", 4, F.TYPE_STRING, OPT, {})])
    msg("GraphProto", [("node", 1, F.TYPE_MESSAGE, REP, {"type_name": ".onnx_subset.NodeProto"}),
                       ("name", 2, F.TYPE_STRING, OPT, {}),
                       ("initializer", 5, F.TYPE_MESSAGE, REP, {"type_name": ".onnx_subset.TensorProto"})])
    msg("ModelProto", [("ir_version", 1, F.TYPE_INT64, OPT, {}), ("producer_name", 2, F.TYPE_STRING, OPT, {}),
                       ("graph", 7, F.TYPE_MESSAGE, OPT, {"type_name": ".onnx_subset.GraphProto"})])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:  # older protobuf
        factory = message_factory.MessageFactory(pool)
        get = factory.GetPrototype
    return {n: get(pool.FindMessageTypeByName("onnx_subset." + n)) for n in ("ModelProto", "GraphProto", "NodeProto", "TensorProto")}


def _conv_order():
    order = ["%s.%s.conv_s1." % (s, l[0]) for s, kind, cin, mid, cout in STAGES for l in rsu_layers(kind, cin, mid, cout)]
    return order + [s + "." for s, _ in SIDES] + ["outconv."]


def _random_state(seed=0):
    rng = np.random.default_rng(seed)
    sd = {}
    for k, shp in param_spec().items():
        n = int(np.prod(shp))
        a = rng.standard_normal(min(n, 4096)).astype(np.float32)       # tile a short random vector: fast, still position-sensitive
        a = np.resize(a, n) + np.float32(len(sd) * 1e-3)
        sd[k] = (np.abs(a) + 0.5 if k.endswith("running_var") else a).reshape(shp).astype(np.float32)
    return sd


def test_wire_reader_against_the_protobuf_runtime():
    cls = _protobuf_classes()
    rng = np.random.default_rng(1)
    m = cls["ModelProto"](ir_version=7, producer_name="pytorch")
    g = m.graph
    g.name = "torch_jit"
    a = rng.standard_normal((4, 3, 3, 3)).astype(np.float32)
    b = rng.standard_normal((4,)).astype(np.float32)
    t = g.initializer.add(name="w", data_type=1, raw_data=a.tobytes())
    t.dims.extend(a.shape)
    t = g.initializer.add(name="b", data_type=1)
    t.dims.extend(b.shape)
    t.float_data.extend(b.tolist())                                   # packed float_data instead of raw_data
    t = g.initializer.add(name="shape_const", data_type=7)            # INT64 constant: must be skipped, not an error
    t.dims.append(2)
    t.int64_data.extend([1, -1])
    g.node.add(op_type="Conv", name="Conv_0", input=["x", "w", "b"], output=["y"])
    g.node.add(op_type="Relu", name="Relu_1", input=["y"], output=["z"])
    nodes, inits = ow.read_graph(m.SerializeToString())
    assert [n["op"] for n in nodes] == ["Conv", "Relu"] and nodes[0]["inputs"] == ["x", "w", "b"] and nodes[1]["outputs"] == ["z"]
    assert set(inits) == {"w", "b"} and np.array_equal(inits["w"], a) and np.array_equal(inits["b"], b)
    # and the module's own writer parses with the official runtime
    blob = ow.encode_model([ow.encode_node("Conv", ["x", "w"], ["y"], "c0")], [ow.encode_tensor("w", a), ow.encode_tensor("b", b, raw=False)])
    back = cls["ModelProto"].FromString(blob)
    assert back.ir_version == 7 and back.graph.node[0].op_type == "Conv" and list(back.graph.initializer[0].dims) == [4, 3, 3, 3]
    assert np.array_equal(np.frombuffer(back.graph.initializer[0].raw_data, np.float32).reshape(a.shape), a)
    assert np.array_equal(np.array(back.graph.initializer[1].float_data, np.float32), b)


def test_u2net_named_initializers():
    sd = _random_state(0)
    tensors = [ow.encode_tensor(k, v) for k, v in sd.items()]
    got = ow.u2net_state_dict(ow.encode_model([], tensors))
    assert set(got) == set(sd) and all(np.array_equal(got[k], sd[k]) for k in sd)


@pytest.mark.parametrize("folded", [False, True])
def test_u2net_anonymous_initializers_in_graph_order(folded):
    sd = _random_state(1)
    nodes, tensors, n = [], [], 0
    cur = "input"
    for p in _conv_order():
        wn, bn_ = "%d" % (1000 + n), "%d" % (1001 + n)
        n += 2
        tensors += [ow.encode_tensor(wn, sd[p + "weight"]), ow.encode_tensor(bn_, sd[p + "bias"])]
        out = "conv_out_%d" % n
        nodes.append(ow.encode_node("Conv", [cur, wn, bn_], [out], "Conv_%d" % n))
        cur = out
        if p.endswith("conv_s1."):
            q = p[:-len("conv_s1.")] + "bn_s1."
            if not folded:
                names = []
                for k in ("weight", "bias", "running_mean", "running_var"):
                    names.append("%d" % (1000 + n))
                    tensors.append(ow.encode_tensor(names[-1], sd[q + k]))
                    n += 1
                out = "bn_out_%d" % n
                nodes.append(ow.encode_node("BatchNormalization", [cur] + names, [out], "BN_%d" % n))
                cur = out
            out = "relu_out_%d" % n
            nodes.append(ow.encode_node("Relu", [cur], [out]))
            cur = out
    got = ow.u2net_state_dict(ow.encode_model(nodes, tensors))
    for k, v in sd.items():
        if folded and ".bn_s1." in k:
            want = {"weight": 1.0, "bias": 0.0, "running_mean": 0.0, "running_var": 1.0 - BN_EPS}[k.rsplit(".", 1)[1]]
            assert np.all(got[k] == np.float32(want)), k
        else:
            assert np.array_equal(got[k], v), k
    if folded:   # the identity BatchNorm folds back to the convolution itself
        k = "stage1.rebnconvin.bn_s1."
        scale = got[k + "weight"] / np.sqrt(got[k + "running_var"] + np.float32(BN_EPS))
        assert np.allclose(scale, 1.0, atol=1e-7)


def test_rejects_a_graph_that_is_not_u2net():
    sd = _random_state(2)
    nodes = [ow.encode_node("Conv", ["x", "w0"], ["y"])]
    with pytest.raises(ow.OnnxFormatError):
        ow.u2net_state_dict(ow.encode_model(nodes, [ow.encode_tensor("w0", sd["side1.weight"])]))
    bad = dict(sd)
    bad["stage3.rebnconv2.conv_s1.weight"] = bad["stage3.rebnconv2.conv_s1.weight"][:, :-1]
    with pytest.raises(ow.OnnxFormatError):
        ow.u2net_state_dict(ow.encode_model([], [ow.encode_tensor(k, v) for k, v in bad.items()]))
    with pytest.raises(ow.OnnxFormatError):
        ow.read_graph(b"\x08\x07")            # a ModelProto without a graph
    with pytest.raises(ow.OnnxFormatError):
        ow.read_graph(ow.encode_model([], [ow.encode_tensor("w", np.ones(4, np.float32))])[:-3])   # truncated file


def test_session_load_weights_dispatches_on_extension(tmp_path):
    from sculptmate_amd.rembg import session

    sd = _random_state(3)
    path = str(tmp_path / "u2net.onnx")
    with open(path, "wb") as fh:
        fh.write(ow.encode_model([], [ow.encode_tensor(k, v) for k, v in sd.items()]))
    got = session.load_weights(path)
    assert all(np.array_equal(got[k], sd[k]) for k in sd)
    with pytest.raises(FileNotFoundError):
        session.load_weights(str(tmp_path / "missing.onnx"))
