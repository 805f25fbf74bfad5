"""Python view of the C-ABI loaders (the reference's on-disk formats, SURVEY.md
Appendix C).  Parsing happens in libopenpbso_amd.so (csrc/loaders.cpp)."""
import ctypes as C

import numpy as np

from . import capi


def read_modes(path):
    """ModeData<double>::read (ModeData.h:61-83) -> (omega_squared[n], modes[n][n_dof])."""
    l = capi.lib()
    nd, nm = C.c_int(0), C.c_int(0)
    om, md = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    rc = l.pbso_modes_read(path.encode(), C.byref(nd), C.byref(nm), C.byref(om), C.byref(md))
    if rc != capi.OK:
        raise IOError(f"cannot open file for reading modes: {path} (status {rc})")
    try:
        omega2 = np.ctypeslib.as_array(om, shape=(max(nm.value, 1),))[: nm.value].copy()
        modes = np.ctypeslib.as_array(md, shape=(max(nm.value * nd.value, 1),))[: nm.value * nd.value].copy()
    finally:
        l.pbso_free(om)
        l.pbso_free(md)
    return omega2, modes.reshape(nm.value, nd.value)


def num_modes_audible(omega_squared, density, audible_freq):
    """ModeData<double>::numModesAudible (ModeData.h:120-148)."""
    om = np.ascontiguousarray(omega_squared, dtype=np.float64)
    return capi.lib().pbso_num_modes_audible(om.ctypes.data_as(C.POINTER(C.c_double)), om.size,
                                             float(density), float(audible_freq))


def read_material(path):
    """ModalMaterial<double>::Read (ModalMaterial.h:35-55) -> dict or None."""
    out = (C.c_double * 5)()
    rc = capi.lib().pbso_material_read(path.encode(), out)
    if rc != capi.OK:
        return None
    return dict(density=out[0], youngsModulus=out[1], poissonRatio=out[2], alpha=out[3], beta=out[4])


def parse_fatcube(data: bytes):
    """FFAT_Map_Serialize_Double::Load (ffat_map_serialize.h:166-254) from bytes."""
    m = capi.FfatMap()
    rc = capi.lib().pbso_fatcube_parse(data, len(data), C.byref(m))
    if rc != capi.OK:
        raise IOError(f"malformed .fatcube (status {rc})")
    try:
        out = dict(
            mode_id=m.mode_id, k=m.k, cell_size=m.cell_size, center3=np.array(m.center3[:]),
            center=np.array(m.center[:]), bbox_low=np.array(m.bbox_low[:]), bbox_top=np.array(m.bbox_top[:]),
            low_corners=np.array([list(r) for r in m.low_corners]),
            n_elements=np.array([list(r) for r in m.n_elements], dtype=np.int32),
            strides=np.array(m.strides[:], dtype=np.int32),
            psi=np.ctypeslib.as_array(m.psi, shape=(max(m.n_psi, 1),))[: m.n_psi].copy())
    finally:
        capi.lib().pbso_ffat_map_free(C.byref(m))
    return out


def read_obj(path):
    """igl::read_triangle_mesh + igl::per_vertex_normals (tools/real_time_modal_sound.cpp:508-509) ->
    (V [n][3], F [m][3] 0-based, VN [n][3] unit, area-weighted)."""
    l = capi.lib()
    nv, nf = C.c_int(0), C.c_int(0)
    v, vn, f = C.POINTER(C.c_double)(), C.POINTER(C.c_double)(), C.POINTER(C.c_int)()
    rc = l.pbso_obj_read(path.encode(), C.byref(nv), C.byref(nf), C.byref(v), C.byref(f), C.byref(vn))
    if rc != capi.OK:
        raise IOError(f"cannot read triangle mesh {path} (status {rc})")
    try:
        V = np.ctypeslib.as_array(v, shape=(max(3 * nv.value, 1),))[: 3 * nv.value].copy().reshape(-1, 3)
        VN = np.ctypeslib.as_array(vn, shape=(max(3 * nv.value, 1),))[: 3 * nv.value].copy().reshape(-1, 3)
        F = np.ctypeslib.as_array(f, shape=(max(3 * nf.value, 1),))[: 3 * nf.value].copy().reshape(-1, 3)
    finally:
        l.pbso_free(v)
        l.pbso_free(f)
        l.pbso_free(vn)
    return V, F, VN
