"""Host-side mirror of the reference's ModalSolver interface over the C ABI.

`Engine` batches many objects (one reference ModalSolver<double> each) on one
GPU; `ModalSolver` is the one-object facade with the reference's method names
(modal_solver.h:100-179) so that tests read like code written against the
reference.  All numerics happen in libopenpbso_amd.so (HIP); nothing here
computes audio.
"""
import collections
import ctypes as C
import os

import numpy as np

from . import capi

POINT_FORCE = capi.POINT_FORCE
GAUSSIAN_FORCE = capi.GAUSSIAN_FORCE
AUTOREGRESSIVE_FORCE = capi.AUTOREGRESSIVE_FORCE


class PbsoError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"[{status}] {text}")
        self.status = status


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


# numpy view of capi.ForceMsg (same layout: checked at import)
FORCE_MSG_DTYPE = np.dtype([("force_type", np.int32), ("gaussian_width_us", np.float64),
                            ("sustained_force_start", np.int32), ("sustained_force_end", np.int32),
                            ("clear_all_forces", np.int32), ("data_kind", np.int32), ("data", np.uintp),
                            ("n_data", np.int32), ("vids", np.int32, 3), ("coords", np.float64, 3),
                            ("vn", np.float64, 3)], align=True)
assert FORCE_MSG_DTYPE.itemsize == C.sizeof(capi.ForceMsg) and all(
    FORCE_MSG_DTYPE.fields[n][1] == getattr(capi.ForceMsg, n).offset for n, _ in capi.ForceMsg._fields_)


class ForceMessage:
    """ForceMessage<double> (modal_solver.h:27-77).

    The modal `data` is either given explicitly (as the GUI thread built it), or
    as a vertex / face hit to be projected onto the mode shapes on the device
    (GetModalForceVertex / GetModalForceFace, tools/real_time_modal_sound.cpp:236-295).
    """

    def __init__(self, data=None, forceType=POINT_FORCE, gaussianWidth=0.0, sustainedForceStart=False,
                 sustainedForceEnd=False, clearAllForces=False, vid=None, vids=None, coords=None, vn=None):
        self.data = None if data is None else np.ascontiguousarray(data, dtype=np.float64)
        self.forceType = forceType
        self.gaussianWidth = float(gaussianWidth)
        self.sustainedForceStart = sustainedForceStart
        self.sustainedForceEnd = sustainedForceEnd
        self.clearAllForces = clearAllForces
        self.vid, self.vids, self.coords, self.vn = vid, vids, coords, vn

    def to_c(self):
        m = capi.ForceMsg()
        m.force_type = int(self.forceType)
        m.gaussian_width_us = self.gaussianWidth
        m.sustained_force_start = int(bool(self.sustainedForceStart))
        m.sustained_force_end = int(bool(self.sustainedForceEnd))
        m.clear_all_forces = int(bool(self.clearAllForces))
        if self.data is not None:
            m.data_kind = capi.DATA_EXPLICIT
            m.data = _dp(self.data)
            m.n_data = self.data.size
        elif self.vid is not None:
            m.data_kind = capi.DATA_VERTEX
            m.vids[0] = int(self.vid)
            for j in range(3):
                m.vn[j] = float(self.vn[j])
        elif self.vids is not None:
            m.data_kind = capi.DATA_FACE
            for j in range(3):
                m.vids[j] = int(self.vids[j])
                m.coords[j] = float(self.coords[j])
                m.vn[j] = float(self.vn[j])
        else:
            m.data_kind = capi.DATA_ZERO
        return m


def _select_from_env():
    """Test / A-B convenience of THIS wrapper (the library reads no such variables): the kernel- and path-selection fields of
    pbso_engine_desc (ABI 4) from the environment, so that a whole pytest / bench run can pin one path.
    PBSO_ENGINE_OPTS="field=value,..." names descriptor fields directly; the older switches map onto them."""
    env = os.environ
    sel = {}
    split = env.get("PBSO_SPLIT")
    if split == "0":                                  # the block kernel K1b for every launch, walked buffer by buffer
        sel.update(bank_kernel=capi.BANK_BLOCK, time_chunks=-1)
    elif split == "2":                                # the pipeline kernel K1p for every launch
        sel.update(bank_kernel=capi.BANK_PIPE)
    if "PBSO_TIME_CHUNKS" in env:
        sel["time_chunks"] = int(env["PBSO_TIME_CHUNKS"])
    if "PBSO_TC_SHAPE" in env:
        sel["time_chunk_shape"] = int(env["PBSO_TC_SHAPE"])
    if env.get("PBSO_DIRECT_HITS") == "0":
        sel["direct_hits"] = -1
    if env.get("PBSO_FORCED_BLOCK") == "0":
        sel["forced_block"] = -1
    if "PBSO_DENSE_LAUNCHES" in env:
        sel["dense_launches"] = {"block": 1, "sample": 2}[env["PBSO_DENSE_LAUNCHES"]]
    if env.get("PBSO_DEVICE_PROFILES") == "0":
        sel["device_profiles"] = -1
    if env.get("PBSO_AR_SERIAL") == "1":
        sel["profile_kernel"] = 2
    elif env.get("PBSO_K2_ROWS") == "0":
        sel["profile_kernel"] = 1
    for name, field in (("PBSO_K2_MARGIN_PCT", "profile_margin_pct"), ("PBSO_TEAM_WAVES", "team_waves"),
                        ("PBSO_PIPE_CONSUMERS", "pipe_consumers"), ("PBSO_SPLIT_MAX_CHUNKS", "pipe_max_teams"),
                        ("PBSO_CHUNK_BUFFERS", "chunk_buffers"), ("PBSO_PLAN_THREADS", "plan_threads"),
                        ("PBSO_PLAN_PIN", "plan_pin")):
        if name in env:
            sel[field] = int(env[name])
    if "PBSO_K2_PRIO" in env:
        sel["profile_priority"] = int(env["PBSO_K2_PRIO"]) + 1
    if "PBSO_TIMING_EVERY" in env:
        n = int(env["PBSO_TIMING_EVERY"])
        sel["timing_every"] = n if n > 0 else -1
    if env.get("PBSO_WARM_COPIES") == "0":
        sel["warm_copies"] = -1
    for item in filter(None, env.get("PBSO_ENGINE_OPTS", "").split(",")):
        k, v = item.split("=")
        sel[k.strip()] = int(v)
    return sel


SELECT_FIELDS = ("bank_kernel", "time_chunks", "direct_hits", "forced_block", "dense_launches", "device_profiles",
                 "profile_kernel", "profile_margin_pct", "profile_priority", "team_waves", "pipe_consumers",
                 "pipe_max_teams", "chunk_buffers", "plan_threads", "plan_pin", "timing_every", "warm_copies", "stream_sync", "latency_path",
                 "time_chunk_shape", "scan_kernel", "fuse_short_launches", "submit_thread")


def fill_engine_desc(d, device, form, qnorm, modes_per_lane, stream, frames_per_buffer, select):
    """a capi.EngineDesc from the wrapper's arguments; returns the kernel / path selection that went into it"""
    d.abi_version = capi.ABI_VERSION
    d.device = device
    d.frames_per_buffer = frames_per_buffer
    d.recurrence_form = form
    d.qnorm_mode = qnorm
    d.modes_per_lane = modes_per_lane
    d.stream = stream
    sel = _select_from_env()
    sel.update(select)
    for k, v in sel.items():
        if k not in SELECT_FIELDS:
            raise TypeError(f"unknown engine option {k!r}")
        setattr(d, k, int(v))
    return sel


def default_form():
    return {"block": capi.FORM_BLOCK, "velocity": capi.FORM_VELOCITY, "direct": capi.FORM_DIRECT,
            "block_bf16": capi.FORM_BLOCK_BF16}[os.environ.get("PBSO_FORM", "block")]


class Engine:
    @classmethod
    def from_handle(cls, handle, n_modes, qnorm, frames_per_buffer=0):
        """a view of an engine somebody else owns (a device group's rank): same methods, close() leaves it alone"""
        self = cls.__new__(cls)
        self._l = capi.lib()
        self._h = C.c_void_p(handle)
        self._owned = False
        self.n_modes = list(n_modes)
        self.B = frames_per_buffer or 513
        self.qnorm_mode = qnorm
        self._last_nb = 0
        self._borrowed = None
        self.select = {}
        return self

    def __init__(self, device=0, form=None, qnorm=capi.QNORM_ALL, modes_per_lane=0,
                 stream=None, frames_per_buffer=0, **select):
        """select: the ABI-4 fields of pbso_engine_desc that pick kernels and paths for THIS engine (bank_kernel,
        time_chunks, direct_hits, ...; include/openpbso_amd.h).  0 / absent = the engine's policy."""
        if form is None:
            # the C ABI's default (a zeroed pbso_engine_desc): the block form with the exact f32 projection.
            # PBSO_FORM=block|block_bf16|velocity|direct lets a whole test / bench run pick the oscillator-bank kernel
            form = default_form()
        self.form = form
        self._l = capi.lib()
        d = capi.EngineDesc()
        self.select = fill_engine_desc(d, device, form, qnorm, modes_per_lane, stream, frames_per_buffer, select)
        self._owned = True
        h = C.c_void_p()
        rc = self._l.pbso_engine_create(C.byref(d), C.byref(h))
        self._h = h
        if rc != capi.OK:
            text = self._l.pbso_last_error(h).decode() if h else "engine_create failed"
            if h:
                self._l.pbso_engine_destroy(h)
            self._h = None
            raise PbsoError(rc, text)
        self.n_modes = []
        self.B = frames_per_buffer or 513
        self.qnorm_mode = qnorm
        self._last_nb = 0
        self._borrowed = None

    # -- plumbing -----------------------------------------------------------
    def _chk(self, rc):
        if rc < 0:
            raise PbsoError(rc, self._l.pbso_last_error(self._h).decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._l.pbso_engine_destroy(self._h)
            self._h = None
            for p in getattr(self, "_pinned", []):
                self._l.pbso_host_free(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- BuildSolver ----------------------------------------------------------
    def add_object(self, omega_squared, density, alpha, beta, n_modes=None, mode_shapes=None):
        om = np.ascontiguousarray(omega_squared, dtype=np.float64)
        d = capi.ObjectDesc()
        d.n_modes = om.size if n_modes is None else int(n_modes)
        d.n_omega = om.size
        d.omega_squared = _dp(om)
        d.density, d.alpha, d.beta = float(density), float(alpha), float(beta)
        if mode_shapes is not None:
            ms = np.ascontiguousarray(mode_shapes, dtype=np.float64)
            assert ms.ndim == 2 and ms.shape[0] >= d.n_modes
            ms = np.ascontiguousarray(ms[: d.n_modes])
            d.n_dof = ms.shape[1]
            d.mode_shapes = _dp(ms)
        oid = C.c_int(-1)
        self._chk(self._l.pbso_add_object(self._h, C.byref(d), C.byref(oid)))
        self.n_modes.append(d.n_modes)
        return oid.value

    def add_object_from_files(self, modes_path, material_path, ffat_dir=None):
        oid, naud = C.c_int(-1), C.c_int(0)
        self._chk(self._l.pbso_add_object_from_files(
            self._h, modes_path.encode(), material_path.encode(),
            None if ffat_dir is None else ffat_dir.encode(), C.byref(oid), C.byref(naud)))
        self.n_modes.append(naud.value)
        return oid.value, naud.value

    def set_ffat_maps(self, obj, maps):
        """maps: list of dicts with the FFAT_Map<double,3> runtime fields."""
        arr = (capi.FfatMap * len(maps))()
        keep = []
        for i, m in enumerate(maps):
            a = arr[i]
            a.mode_id = int(m["mode_id"])
            a.k = float(m["k"])
            a.cell_size = float(m["cell_size"])
            for j in range(3):
                a.center3[j] = float(m["center3"][j])
                a.center[j] = float(m["center"][j])
                a.bbox_low[j] = float(m["bbox_low"][j])
                a.bbox_top[j] = float(m["bbox_top"][j])
            for f in range(6):
                for j in range(3):
                    a.low_corners[f][j] = float(m["low_corners"][f][j])
                a.n_elements[f][0] = int(m["n_elements"][f][0])
                a.n_elements[f][1] = int(m["n_elements"][f][1])
                a.strides[f] = int(m["strides"][f])
            psi = np.ascontiguousarray(m["psi"], dtype=np.float64)
            keep.append(psi)
            a.n_psi = psi.size
            a.psi = _dp(psi)
        self._chk(self._l.pbso_object_set_ffat_maps(self._h, obj, arr, len(maps)))

    def read_ffat_maps(self, obj, directory):
        self._chk(self._l.pbso_object_read_ffat_maps(self._h, obj, directory.encode()))

    def finalize(self):
        self._chk(self._l.pbso_finalize(self._h))

    # -- messages -------------------------------------------------------------
    def enqueue_force(self, obj, msg, not_before=0):
        return bool(self._chk(self._l.pbso_enqueue_force(self._h, obj, C.byref(msg.to_c()), not_before)))

    @staticmethod
    def hit_messages(objs, vids, vns, not_before, coords=None, force_type=capi.POINT_FORCE):
        """(object ids, pbso_force_msg array, stamps) ready for enqueue_force_batch.  Vertex hits
        (GetModalForceVertex): vids [n]; face hits (GetModalForceFace): vids [n][3] and barycentric
        coords [n][3]; vns [n][3]; objs / not_before int arrays [n]."""
        objs = np.ascontiguousarray(objs, dtype=np.int32)
        stamps = np.ascontiguousarray(not_before, dtype=np.int64)
        n = objs.size
        msgs = np.zeros(n, dtype=FORCE_MSG_DTYPE)
        msgs["force_type"] = force_type
        if coords is None:
            msgs["data_kind"] = capi.DATA_VERTEX
            msgs["vids"][:, 0] = vids
        else:
            msgs["data_kind"] = capi.DATA_FACE
            msgs["vids"] = np.asarray(vids).reshape(n, 3)
            msgs["coords"] = np.asarray(coords, dtype=np.float64).reshape(n, 3)
        msgs["vn"] = np.asarray(vns, dtype=np.float64).reshape(n, 3)
        return objs, msgs, stamps

    def enqueue_force_batch(self, objs, msgs, not_before):
        """pbso_enqueue_force_batch: one call for a whole step of the force script; returns the
        number of messages the queues took."""
        return self._chk(self._l.pbso_enqueue_force_batch(
            self._h, objs.size, objs.ctypes.data_as(C.POINTER(C.c_int)), msgs.ctypes.data_as(C.POINTER(capi.ForceMsg)),
            not_before.ctypes.data_as(C.POINTER(C.c_int64)), None))

    def enqueue_vertex_hits(self, objs, vids, vns, not_before):
        """pbso_enqueue_vertex_hits: a step's plain PointForce vertex hits as parallel arrays, object by object (ids ascending,
        stamps ascending within an object).  The arrays are borrowed by the engine until the next step() returns: they are kept
        alive here; do not modify them in between.  Arrays of the right dtype and layout are passed through without a copy."""
        o = np.ascontiguousarray(objs, dtype=np.int32)
        v = np.ascontiguousarray(vids, dtype=np.int32)
        n = np.ascontiguousarray(vns, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(not_before, dtype=np.int64)
        assert o.size == v.size == t.size == n.shape[0]
        rc = self._chk(self._l.pbso_enqueue_vertex_hits(
            self._h, o.size, o.ctypes.data_as(C.POINTER(C.c_int)), v.ctypes.data_as(C.POINTER(C.c_int)), _dp(n),
            t.ctypes.data_as(C.POINTER(C.c_int64))))
        # (only a script the engine TOOK is held here: a refused call -- a script already pending, bad ids -- leaves the
        #  arrays of the pending one, which the engine still points at, alive)
        self._borrowed = (o, v, n, t)
        return rc

    @staticmethod
    def prepare_vertex_hits(objs, vids, vns, not_before):
        """the arguments of enqueue_vertex_hits converted once (a caller that replays pre-built scripts step after step: the
        conversions cost more than the call for a small step); hand the result to enqueue_prepared_vertex_hits"""
        o = np.ascontiguousarray(objs, dtype=np.int32)
        v = np.ascontiguousarray(vids, dtype=np.int32)
        n = np.ascontiguousarray(vns, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(not_before, dtype=np.int64)
        assert o.size == v.size == t.size == n.shape[0]
        return (o.size, o.ctypes.data_as(C.POINTER(C.c_int)), v.ctypes.data_as(C.POINTER(C.c_int)), _dp(n),
                t.ctypes.data_as(C.POINTER(C.c_int64)), (o, v, n, t))

    def enqueue_prepared_vertex_hits(self, prepared):
        rc = self._chk(self._l.pbso_enqueue_vertex_hits(self._h, *prepared[:5]))
        self._borrowed = prepared[5]
        return rc

    def enqueue_arprm(self, obj, a, sigma, mu, not_before=0):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return bool(self._chk(self._l.pbso_enqueue_arprm(self._h, obj, _dp(a), sigma, mu, not_before)))

    def arprm_pending(self, obj):
        """pbso_arprm_pending: True while the object's AR-parameter slot is taken (a try_enqueue would fail, modal_solver.h:378-381)"""
        return bool(self._chk(self._l.pbso_arprm_pending(self._h, obj)))

    def compute_transfer(self, obj, pos, not_before=0):
        p = np.ascontiguousarray(pos, dtype=np.float64)
        return bool(self._chk(self._l.pbso_compute_transfer(self._h, obj, _dp(p), not_before)))

    def compute_transfer_path(self, objs, pos, not_before):
        """pbso_compute_transfer_path: n computeTransfer(pos) calls in one; returns the accepted flags [n]"""
        o = np.ascontiguousarray(objs, dtype=np.int32)
        p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(not_before, dtype=np.int64)
        assert o.size == t.size == p.shape[0]
        acc = np.zeros(o.size, dtype=np.uint8)
        self._chk(self._l.pbso_compute_transfer_path(self._h, o.size, o.ctypes.data_as(C.POINTER(C.c_int)), _dp(p),
                                                     t.ctypes.data_as(C.POINTER(C.c_int64)), acc.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return acc.astype(bool)

    def listeners_enable(self, obj):
        """from the next step on, keep this object's block-start states for mix_listeners"""
        self._chk(self._l.pbso_listeners_enable(self._h, obj))

    def mix_listeners(self, obj, pos):
        """the last step's audio of `obj` at every listener position: [n_listeners][n_buffers * 513] float32"""
        p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        out = np.empty((p.shape[0], self._last_nb * self.B), dtype=np.float32)
        self._chk(self._l.pbso_mix_listeners(self._h, obj, _dp(p), p.shape[0], out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out

    def n_maps(self, obj):
        """_ffat_maps->size() (0 before readFFATMaps)"""
        return self._chk(self._l.pbso_object_n_maps(self._h, obj))

    def compute_transfer_batch(self, obj, pos, n_cols=None, out=None):
        """computeTransfer(pos, T*) for many positions: [n_pos][n_cols]; columns beyond the object's map count stay 0.
        out: a C-contiguous float64 array [n_pos][n_cols] to fill (a caller that asks every frame keeps one: a fresh
        84 MB array costs more in page faults than the lookups and the copy together)"""
        p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        if n_cols is None:
            n_cols = self.n_maps(obj) if out is None else out.shape[1]
        if out is None:
            out = np.zeros((p.shape[0], n_cols))
        assert out.dtype == np.float64 and out.flags.c_contiguous and out.shape == (p.shape[0], n_cols)
        rc = self._chk(self._l.pbso_compute_transfer_batch(self._h, obj, _dp(p), p.shape[0], _dp(out), n_cols))
        return bool(rc), out

    def set_use_transfer(self, obj, use, not_before=0):
        self._chk(self._l.pbso_set_use_transfer(self._h, obj, int(use), not_before))

    def latest_transfer(self, obj):
        out = np.empty(max(self.n_modes[obj], 1))
        self._chk(self._l.pbso_get_latest_transfer(self._h, obj, _dp(out)))
        return out[: self.n_modes[obj]]

    # -- stepping -------------------------------------------------------------
    def step(self, n_buffers=1, into=None):
        if into is None:
            self._chk(self._l.pbso_step(self._h, n_buffers))
        else:
            self._chk(self._l.pbso_step_into(self._h, n_buffers, C.c_void_p(into)))
        self._last_nb = n_buffers
        self._borrowed = None                           # (a hit script is consumed by the step that follows it)

    def sync(self):
        self._chk(self._l.pbso_sync(self._h))

    def flush(self):
        """pbso_flush: with submit_thread, until the recorded launches of the steps so far are in their streams (not a device sync)"""
        self._chk(self._l.pbso_flush(self._h))

    def host_buffer(self, n_buffers):
        """a pinned numpy array [n_objects][n_buffers * 513] float32 for step_to_host (freed with the engine)"""
        n = len(self.n_modes) * n_buffers * self.B
        p = C.c_void_p()
        rc = self._l.pbso_host_alloc(n * 4, C.byref(p))
        if rc != capi.OK:
            raise PbsoError(rc, "pbso_host_alloc failed")
        if not hasattr(self, "_pinned"):
            self._pinned = []
        self._pinned.append(p)
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n,))
        return arr.reshape(len(self.n_modes), n_buffers * self.B)

    def step_to_host(self, n_buffers, out):
        """pbso_step_to_host: the step's audio of all objects into `out` (a host_buffer), asynchronously"""
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size == len(self.n_modes) * n_buffers * self.B
        self._chk(self._l.pbso_step_to_host(self._h, n_buffers, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        self._last_nb = n_buffers
        self._borrowed = None

    def host_wait(self):
        self._chk(self._l.pbso_host_wait(self._h))

    def audio(self):
        n = len(self.n_modes) * self._last_nb * self.B
        out = np.empty(n, dtype=np.float32)
        self._chk(self._l.pbso_read_audio(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), n))
        return out.reshape(len(self.n_modes), self._last_nb * self.B)

    def audio_rows(self, rows):
        """the last step's audio of some objects only: [len(rows)][n_buffers * 513] float32"""
        r = np.ascontiguousarray(rows, dtype=np.int32)
        out = np.empty((r.size, self._last_nb * self.B), dtype=np.float32)
        self._chk(self._l.pbso_read_audio_rows(self._h, r.ctypes.data_as(C.POINTER(C.c_int)), r.size, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def emitted(self):
        n = len(self.n_modes) * self._last_nb
        out = np.empty(n, dtype=np.uint8)
        self._chk(self._l.pbso_read_emitted(self._h, out.ctypes.data_as(C.POINTER(C.c_ubyte)), n))
        return out.reshape(len(self.n_modes), self._last_nb).astype(bool)

    def qnorm(self, obj, buffer):
        n = self.n_modes[obj]
        out = np.empty(max(n, 1), dtype=np.float32)
        self._chk(self._l.pbso_read_qnorm(self._h, obj, buffer, out.ctypes.data_as(C.POINTER(C.c_float)), n))
        return out[:n]

    def state(self, obj):
        n = self.n_modes[obj]
        q1, q2 = np.empty(max(n, 1)), np.empty(max(n, 1))
        self._chk(self._l.pbso_read_state(self._h, obj, _dp(q1), _dp(q2), n))
        return q1[:n], q2[:n]

    def set_state(self, obj, q1, q2):
        """pbso_write_state: restore (q_{k-1}, q_{k-2}) as state() returned them"""
        a = np.ascontiguousarray(q1, dtype=np.float64)
        b = np.ascontiguousarray(q2, dtype=np.float64)
        assert a.size == b.size
        self._chk(self._l.pbso_write_state(self._h, obj, _dp(a), _dp(b), a.size))

    def census(self, n_teams=None):
        """per-workgroup (start, end [100 MHz ticks], HW_ID, XCC_ID, clock start/end, block form: cycles in head / pipeline / barrier / combine) of the last launch (PBSO_CENSUS=1).
        n_teams: the number of rows to read when the launch ran on the kernel of under-filled scenes (one team per 64 modes)."""
        n = (self.info()["n_teams"] if n_teams is None else n_teams) * 12
        out = np.empty(n, dtype=np.uint64)
        self._chk(self._l.pbso_read_census(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), n))
        return out.reshape(-1, 12)

    def audio_device_ptr(self):
        return self._l.pbso_audio_device_ptr(self._h)

    def mix_objects(self, d_out):
        """pbso_mix_objects: the last step's audio summed over the objects, into the device buffer d_out [n_buffers * 513] f32"""
        self._chk(self._l.pbso_mix_objects(self._h, C.c_void_p(d_out)))

    def info(self):
        i = capi.EngineInfo()
        self._chk(self._l.pbso_get_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in capi.EngineInfo._fields_}


class ModalSolver:
    """One-object facade with the reference's ModalSolver<double> method names
    (modal_solver.h:100-179).  step() produces one 513-sample buffer."""

    def __init__(self, omega_squared, density, alpha, beta, N_modes=None, mode_shapes=None, **engine_kw):
        self.engine = Engine(**engine_kw)
        self.obj = self.engine.add_object(omega_squared, density, alpha, beta, N_modes, mode_shapes)
        self._N_modes = self.engine.n_modes[self.obj]
        self._queue_sound = collections.deque()       # ReaderWriterQueue(2): 3 usable slots
        self._queue_qnorm = collections.deque()
        self._pending_maps = None
        self._final = False
        self._host = None                             # pinned [1][513]: the bank stores the buffer's samples straight into it

    def readFFATMaps(self, maps_or_dir):
        if isinstance(maps_or_dir, str):
            self.engine.read_ffat_maps(self.obj, maps_or_dir)
        else:
            self.engine.set_ffat_maps(self.obj, maps_or_dir)

    def _ensure(self):
        if not self._final:
            self.engine.finalize()
            self._final = True

    def enqueueForceMessage(self, mess):
        self._ensure()
        return self.engine.enqueue_force(self.obj, mess)

    def enqueueForceMessageNoFail(self, mess, maxIte=-1):
        return self.enqueueForceMessage(mess)

    def enqueueArprmMessageNoFail(self, a, sigma, mu, maxIte=-1):
        self._ensure()
        return self.engine.enqueue_arprm(self.obj, a, sigma, mu)

    def computeTransfer(self, pos, out=None):
        self._ensure()
        if out is None:
            return self.engine.compute_transfer(self.obj, pos)
        ok, vals = self.engine.compute_transfer_batch(self.obj, pos, out.size)
        if ok:
            out[:] = vals[0]
        return ok

    def setUseTransfer(self, s):
        self._ensure()
        self.engine.set_use_transfer(self.obj, s)

    def getLatestTransfer(self):
        self._ensure()
        return self.engine.latest_transfer(self.obj)

    def step(self):
        self._ensure()
        if self._host is None:
            self._host = self.engine.host_buffer(1)
        # the buffer's samples arrive in pinned host memory with the kernel's own stores (pbso_step_to_host): no copy call
        # between the oscillator bank and the SoundMessage
        self.engine.step_to_host(1, self._host)
        self.engine.host_wait()
        if not self.engine.emitted()[0, 0]:
            return                                   # clearAllForces: no SoundMessage (modal_solver.h:186-189)
        if self.engine.qnorm_mode != capi.QNORM_OFF and len(self._queue_qnorm) < 3:
            self._queue_qnorm.append(self.engine.qnorm(0, 0))     # try_enqueue, may drop (:273)
        self._queue_sound.append(self._host[0].copy())            # enqueueSoundMessageNoFail (:275)

    def dequeueSoundMessage(self):
        if not self._queue_sound:
            return None
        return self._queue_sound.popleft()

    def getQBufferNorm(self):
        if self._queue_qnorm:
            return self._queue_qnorm.popleft()
        return np.zeros(self._N_modes, dtype=np.float32)
