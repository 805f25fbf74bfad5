"""Encoder for .fatcube files with the Python protobuf runtime: the message
classes are built at run time from a descriptor written after ffat_map.proto
(no protoc in the image).  Used to make fixtures and test directories."""
import numpy as np  # noqa: F401


def proto_classes():
    """Message classes for ffat_map.proto built at run time (no protoc here)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fdp = descriptor_pb2.FileDescriptorProto(name="ffat_map.proto", package="ffat_map", syntax="proto3")

    def msg(name, fields):
        m = fdp.message_type.add(name=name)
        for fname, num, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = ".ffat_map." + tname
    REP, OPT = F.LABEL_REPEATED, F.LABEL_OPTIONAL
    msg("vec", [("item", 1, F.TYPE_DOUBLE, REP, None)])
    msg("mat", [("item", 1, F.TYPE_MESSAGE, REP, "vec")])
    msg("vec_i", [("item", 1, F.TYPE_INT32, REP, None)])
    msg("mat_i", [("item", 1, F.TYPE_MESSAGE, REP, "vec_i")])
    msg("ffat_map_t_1", [("cellsize", 1, F.TYPE_DOUBLE, OPT, None), ("lowcorners", 2, F.TYPE_MESSAGE, OPT, "mat"),
                         ("n_elements", 3, F.TYPE_MESSAGE, OPT, "mat_i"), ("strides", 4, F.TYPE_MESSAGE, OPT, "vec_i"),
                         ("center", 5, F.TYPE_MESSAGE, OPT, "vec"), ("bboxlow", 6, F.TYPE_MESSAGE, OPT, "vec"),
                         ("bboxtop", 7, F.TYPE_MESSAGE, OPT, "vec")])
    msg("ffat_map_t_3", [("k", 1, F.TYPE_DOUBLE, OPT, None), ("center", 2, F.TYPE_MESSAGE, OPT, "vec"),
                         ("shells", 3, F.TYPE_MESSAGE, OPT, "ffat_map_t_1"), ("is_compressed", 4, F.TYPE_BOOL, OPT, None),
                         ("psi", 5, F.TYPE_MESSAGE, OPT, "mat"), ("modeid", 6, F.TYPE_INT32, OPT, None)])
    msg("ffat_map_double", [("map", 1, F.TYPE_MESSAGE, OPT, "ffat_map_t_3")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("ffat_map.ffat_map_double"))


def encode_fatcube(cls, m):
    """Mirrors FFAT_Map_Serialize_Double::Save's field use (ffat_map_serialize.h:90-164)."""
    top = cls()
    m3 = top.map
    m3.k = float(m["k"])
    m3.center.item.extend([float(x) for x in m["center3"]])
    sh = m3.shells
    sh.cellsize = float(m["cell_size"])
    for row in m["low_corners"]:
        sh.lowcorners.item.add().item.extend([float(x) for x in row])
    for row in m["n_elements"]:
        sh.n_elements.item.add().item.extend([int(x) for x in row])
    sh.strides.item.extend([int(x) for x in m["strides"]])
    sh.center.item.extend([float(x) for x in m["center"]])
    sh.bboxlow.item.extend([float(x) for x in m["bbox_low"]])
    sh.bboxtop.item.extend([float(x) for x in m["bbox_top"]])
    m3.is_compressed = False
    m3.psi.item.add().item.extend([float(x) for x in m["psi"]])      # one column
    m3.modeid = int(m["mode_id"])
    return top.SerializeToString()


