"""ctypes binding of oracle/libpbso_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product package (openpbso_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
B = 513
SAMPLE_RATE = 44100
POINT, GAUSSIAN, AR = 0, 1, 2


class OrRng(C.Structure):
    _fields_ = [("x", C.c_uint32), ("saved", C.c_double), ("saved_available", C.c_int)]


class OrForce(C.Structure):
    _fields_ = [
        ("type", C.c_int), ("used", C.c_int), ("width", C.c_double),
        ("width_samples", C.c_int), ("count", C.c_int), ("center", C.c_int),
        ("cutoff", C.c_int), ("buf", C.c_double * 3), ("buf_idx", C.c_int),
        ("a", C.c_double * 2), ("sigma", C.c_double), ("mu", C.c_double),
        ("rng", OrRng),
    ]


class OrFfatMap(C.Structure):
    _fields_ = [
        ("mode_id", C.c_int), ("k", C.c_double), ("center3", C.c_double * 3),
        ("is_compressed", C.c_int), ("cell_size", C.c_double),
        ("low_corners", (C.c_double * 3) * 6), ("n_elements", (C.c_int * 2) * 6),
        ("strides", C.c_int * 6), ("center", C.c_double * 3),
        ("bbox_low", C.c_double * 3), ("bbox_top", C.c_double * 3),
        ("n_psi", C.c_int), ("psi", C.POINTER(C.c_double)),
    ]


def build(native=False):
    """(Re)build the oracle library with gcc. Returns the .so path."""
    target = "native" if native else "libpbso_oracle.so"
    subprocess.run(["make", "-C", _HERE, target], check=True, capture_output=True)
    return os.path.join(_HERE, "libpbso_oracle_native.so" if native else "libpbso_oracle.so")


_lib = None


def lib(native=False):
    global _lib
    if native:
        return _bind(C.CDLL(build(native=True)))
    if _lib is None:
        path = os.path.join(_HERE, "libpbso_oracle.so")
        src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("pbso_oracle.c", "pbso_oracle.h"))
        if not os.path.exists(path) or os.path.getmtime(path) < src_m:
            build()
        try:
            _lib = _bind(C.CDLL(path))
        except OSError:
            build()
            _lib = _bind(C.CDLL(path))
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _bind(l):
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    l.or_build_ab.argtypes = [C.c_double, dp, C.c_int, C.c_double, C.c_double, dp, dp]
    l.or_iir_coeffs.argtypes = [dp, dp, C.c_int, C.c_double, dp, dp, dp]
    l.or_integrator_build.restype = vp
    l.or_integrator_build.argtypes = [C.c_double, dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int]
    l.or_integrator_free.argtypes = [vp]
    l.or_integrator_step.restype = dp
    l.or_integrator_step.argtypes = [vp, dp]
    l.or_integrator_step_free.restype = dp
    l.or_integrator_step_free.argtypes = [vp]
    l.or_rng_init.argtypes = [C.POINTER(OrRng)]
    l.or_rng_normal.restype = C.c_double
    l.or_rng_normal.argtypes = [C.POINTER(OrRng)]
    fp = C.POINTER(OrForce)
    l.or_force_init_point.argtypes = [fp]
    l.or_force_init_gaussian.argtypes = [fp, C.c_double]
    l.or_force_init_ar.argtypes = [fp]
    l.or_force_ar_set_param.argtypes = [fp, dp, C.c_double, C.c_double]
    l.or_force_add.restype = C.c_int
    l.or_force_add.argtypes = [fp, dp]
    l.or_modal_force_vertex.argtypes = [C.c_int, dp, C.c_int, C.c_int, dp, dp]
    l.or_modal_force_face.argtypes = [C.c_int, dp, C.c_int, ip, dp, dp, dp]
    mp = C.POINTER(OrFfatMap)
    l.or_ffat_intersect.argtypes = [mp, dp, dp, ip]
    l.or_ffat_interpolate.argtypes = [mp, dp, ip, ip, dp]
    l.or_ffat_quad_stride.restype = C.c_int
    l.or_ffat_quad_stride.argtypes = [mp, ip]
    l.or_ffat_get_map_val.restype = C.c_double
    l.or_ffat_get_map_val.argtypes = [mp, dp]
    l.or_ffat_make_uniform_cube.argtypes = [mp, C.c_int, C.c_double, dp, C.c_double, C.c_int, dp]
    l.or_ffat_free.argtypes = [mp]
    l.or_modes_read.restype = C.c_int
    l.or_modes_read.argtypes = [C.c_char_p, ip, ip, C.POINTER(dp), C.POINTER(dp)]
    l.or_modes_write.restype = C.c_int
    l.or_modes_write.argtypes = [C.c_char_p, C.c_int, C.c_int, dp, dp]
    l.or_num_modes_audible.restype = C.c_int
    l.or_num_modes_audible.argtypes = [dp, C.c_int, C.c_double, C.c_double]
    l.or_material_read.restype = C.c_int
    l.or_material_read.argtypes = [C.c_char_p, dp]
    l.or_fatcube_parse.restype = C.c_int
    l.or_fatcube_parse.argtypes = [C.c_char_p, C.c_size_t, mp]
    l.or_fatcube_load.restype = C.c_int
    l.or_fatcube_load.argtypes = [C.c_char_p, mp]
    l.or_solver_new.restype = vp
    l.or_solver_new.argtypes = [C.c_int]
    l.or_solver_free.argtypes = [vp]
    l.or_solver_set_integrator.argtypes = [vp, vp]
    l.or_solver_set_ffat_maps.argtypes = [vp, mp, C.c_int]
    l.or_solver_enqueue_force.restype = C.c_int
    l.or_solver_enqueue_force.argtypes = [vp, dp, C.c_int, fp, C.c_int, C.c_int, C.c_int]
    l.or_solver_enqueue_arprm.restype = C.c_int
    l.or_solver_enqueue_arprm.argtypes = [vp, dp, C.c_double, C.c_double]
    l.or_solver_compute_transfer.restype = C.c_int
    l.or_solver_compute_transfer.argtypes = [vp, dp]
    l.or_solver_compute_transfer_out.restype = C.c_int
    l.or_solver_compute_transfer_out.argtypes = [vp, dp, dp]
    l.or_solver_set_use_transfer.argtypes = [vp, C.c_int]
    l.or_solver_latest_transfer.restype = dp
    l.or_solver_latest_transfer.argtypes = [vp]
    l.or_solver_step.restype = C.c_int
    l.or_solver_step.argtypes = [vp, dp, dp]
    l.or_solver_n_active.restype = C.c_int
    l.or_solver_n_active.argtypes = [vp]
    l.or_solver_state.restype = dp
    l.or_solver_state.argtypes = [vp, C.c_int]
    l.or_pa_callback_convert.argtypes = [dp, C.c_int, C.POINTER(C.c_float)]
    l.or_bench_run.restype = C.c_double
    l.or_bench_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, C.c_double, C.c_double,
                               C.c_double, dp, C.c_char_p, dp, C.c_int]
    return l


# --------------------------------------------------------------------------- #
# thin numpy-level helpers
# --------------------------------------------------------------------------- #
def iir_coeffs(omega2, density, alpha, beta, h=1.0 / SAMPLE_RATE):
    """modal_integrator.h:47-101 -> (c1, c2, c3) float64 arrays."""
    om = np.ascontiguousarray(omega2, dtype=np.float64)
    n = om.size
    a, b = np.empty(n), np.empty(n)
    c1, c2, c3 = np.empty(n), np.empty(n), np.empty(n)
    lib().or_build_ab(density, _dp(om), n, alpha, beta, _dp(a), _dp(b))
    lib().or_iir_coeffs(_dp(a), _dp(b), n, h, _dp(c1), _dp(c2), _dp(c3))
    return c1, c2, c3


def make_force(kind, width_us=0.0):
    f = OrForce()
    if kind == POINT:
        lib().or_force_init_point(C.byref(f))
    elif kind == GAUSSIAN:
        lib().or_force_init_gaussian(C.byref(f), float(width_us))
    elif kind == AR:
        lib().or_force_init_ar(C.byref(f))
    else:
        raise ValueError(kind)
    return f


def force_add(f, buf):
    assert buf.dtype == np.float64 and buf.size == B
    return bool(lib().or_force_add(C.byref(f), _dp(buf)))


def modal_force_vertex(modes, vid, vn, n=None):
    modes = np.ascontiguousarray(modes, dtype=np.float64)
    n = modes.shape[0] if n is None else n
    out = np.empty(n)
    vn = np.ascontiguousarray(vn, dtype=np.float64)
    lib().or_modal_force_vertex(n, _dp(modes), modes.shape[1], int(vid), _dp(vn), _dp(out))
    return out


def modal_force_face(modes, vids, coords, vn, n=None):
    modes = np.ascontiguousarray(modes, dtype=np.float64)
    n = modes.shape[0] if n is None else n
    out = np.empty(n)
    vids = np.ascontiguousarray(vids, dtype=np.int32)
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    vn = np.ascontiguousarray(vn, dtype=np.float64)
    lib().or_modal_force_face(n, _dp(modes), modes.shape[1],
                              vids.ctypes.data_as(C.POINTER(C.c_int)), _dp(coords), _dp(vn), _dp(out))
    return out


def uniform_cube(mode_id, k, center, cell_size, dim, psi):
    m = OrFfatMap()
    center = np.ascontiguousarray(center, dtype=np.float64)
    psi = np.ascontiguousarray(psi, dtype=np.float64)
    assert psi.size == 6 * dim * dim
    lib().or_ffat_make_uniform_cube(C.byref(m), mode_id, float(k), _dp(center), float(cell_size), dim, _dp(psi))
    return m


def ffat_get_map_val(m, p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    return lib().or_ffat_get_map_val(C.byref(m), _dp(p))


def fatcube_parse(data: bytes):
    m = OrFfatMap()
    rc = lib().or_fatcube_parse(data, len(data), C.byref(m))
    return rc, m


def map_psi(m):
    return np.ctypeslib.as_array(m.psi, shape=(m.n_psi,)).copy()


class Solver:
    """Mirror of ModalSolver<double> (modal_solver.h:100-179) over the C oracle."""

    def __init__(self, omega2, density, alpha, beta, n_modes=None, h=1.0 / SAMPLE_RATE):
        om = np.ascontiguousarray(omega2, dtype=np.float64)
        self.n = om.size if n_modes is None else n_modes
        self._s = lib().or_solver_new(self.n)
        it = lib().or_integrator_build(density, _dp(om), om.size, alpha, beta, h, self.n)
        assert it, "N for modal integrator invalid"
        lib().or_solver_set_integrator(self._s, it)
        self._maps = None

    def close(self):
        if self._s:
            lib().or_solver_free(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def read_ffat_maps(self, maps):
        arr = (OrFfatMap * len(maps))(*maps)
        lib().or_solver_set_ffat_maps(self._s, arr, len(maps))

    def enqueue_force(self, data, force=None, sustained_start=False, sustained_end=False, clear_all=False):
        d = np.ascontiguousarray(data, dtype=np.float64)
        fp = C.byref(force) if force is not None else None
        return bool(lib().or_solver_enqueue_force(self._s, _dp(d), d.size, fp, int(sustained_start),
                                                  int(sustained_end), int(clear_all)))

    def enqueue_arprm(self, a, sigma, mu):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return bool(lib().or_solver_enqueue_arprm(self._s, _dp(a), sigma, mu))

    def compute_transfer(self, pos):
        p = np.ascontiguousarray(pos, dtype=np.float64)
        return lib().or_solver_compute_transfer(self._s, _dp(p))

    def compute_transfer_out(self, pos, n):
        p = np.ascontiguousarray(pos, dtype=np.float64)
        out = np.empty(n)
        rc = lib().or_solver_compute_transfer_out(self._s, _dp(p), _dp(out))
        return rc, out

    def set_use_transfer(self, use):
        lib().or_solver_set_use_transfer(self._s, int(use))

    def latest_transfer(self):
        return np.ctypeslib.as_array(lib().or_solver_latest_transfer(self._s), shape=(self.n,)).copy()

    def step(self):
        """returns (sound[513], qnorm[n]) or None when the step returned early."""
        sound = np.empty(B)
        qn = np.empty(max(self.n, 1))
        rc = lib().or_solver_step(self._s, _dp(sound), _dp(qn))
        if rc < 0:
            raise AssertionError("reference assert would fire (sustained force list size != 1)")
        if rc == 0:
            return None
        return sound, qn[: self.n]

    def n_active(self):
        return lib().or_solver_n_active(self._s)

    def state(self):
        q1 = np.ctypeslib.as_array(lib().or_solver_state(self._s, 0), shape=(self.n,)).copy()
        q2 = np.ctypeslib.as_array(lib().or_solver_state(self._s, 1), shape=(self.n,)).copy()
        return q1, q2
