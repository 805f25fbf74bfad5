"""ctypes view of the device group (include/openpbso_amd.h "device group"): one engine per GPU, the job's objects sharded
over the ranks by the sum of their modes, RCCL called from the C++ library to gather finished audio.  Nothing here
computes; `Group.engine(rank)` hands out the usual Engine wrapper for a local rank's messages."""
import ctypes as C

import numpy as np

from . import capi
from .solver import Engine, PbsoError, default_form, fill_engine_desc, _dp


def unique_id():
    """PBSO_GROUP_ID_BYTES bytes for a job of several processes: ONE process asks, the launcher hands them to all"""
    buf = (C.c_char * capi.GROUP_ID_BYTES)()
    rc = capi.lib().pbso_group_unique_id(buf)
    if rc != capi.OK:
        raise PbsoError(rc, "pbso_group_unique_id failed (librccl not loadable?)")
    return bytes(buf)


class Group:
    def __init__(self, devices, world_size=0, first_rank=0, unique_id=None, form=None, qnorm=capi.QNORM_ALL, modes_per_lane=0,
                 frames_per_buffer=0, transport=capi.GROUP_RCCL, **select):
        self._l = capi.lib()
        self.devices = list(devices)
        self.world = world_size or len(self.devices)
        self.first = first_rank
        self.qnorm_mode = qnorm
        self.B = frames_per_buffer or 513
        d = capi.GroupDesc()
        d.abi_version = capi.ABI_VERSION
        self._dev = (C.c_int * len(self.devices))(*self.devices)
        d.devices = self._dev
        d.n_devices = len(self.devices)
        d.world_size = world_size
        d.first_rank = first_rank
        d.transport = transport
        self._id = C.create_string_buffer(unique_id, capi.GROUP_ID_BYTES) if unique_id is not None else None
        d.unique_id = C.cast(self._id, C.c_void_p) if self._id is not None else None
        fill_engine_desc(d.engine, 0, default_form() if form is None else form, qnorm, modes_per_lane, None, frames_per_buffer, select)
        h = C.c_void_p()
        rc = self._l.pbso_group_create(C.byref(d), C.byref(h))
        self._h = h
        if rc != capi.OK:
            text = self._l.pbso_group_last_error(h).decode() if h else "group_create failed"
            if h:
                self._l.pbso_group_destroy(h)
            self._h = None
            raise PbsoError(rc, text)
        self._modes = None
        self._engines = {}
        self._last_nb = 0

    def _chk(self, rc):
        if rc < 0:
            raise PbsoError(rc, self._l.pbso_group_last_error(self._h).decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            for e in self._engines.values():
                e.close()
            self._l.pbso_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the job ------------------------------------------------------------------
    def plan(self, modes_per_object):
        m = np.ascontiguousarray(modes_per_object, dtype=np.int32)
        self._chk(self._l.pbso_group_plan(self._h, m.ctypes.data_as(C.POINTER(C.c_int)), m.size))
        self._modes = m

    def span(self, rank):
        lo, hi = C.c_int(0), C.c_int(0)
        self._chk(self._l.pbso_group_rank_span(self._h, rank, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def owner(self, global_id):
        r, l = C.c_int(0), C.c_int(0)
        self._chk(self._l.pbso_group_owner(self._h, global_id, C.byref(r), C.byref(l)))
        return r.value, l.value

    def local_ranks(self):
        return range(self.first, self.first + len(self.devices))

    def add_object(self, global_id, omega_squared, density, alpha, beta, mode_shapes=None):
        om = np.ascontiguousarray(omega_squared, dtype=np.float64)
        d = capi.ObjectDesc()
        d.n_modes = int(self._modes[global_id])
        d.n_omega = om.size
        d.omega_squared = _dp(om)
        d.density, d.alpha, d.beta = float(density), float(alpha), float(beta)
        if mode_shapes is not None:
            ms = np.ascontiguousarray(np.asarray(mode_shapes, dtype=np.float64)[: d.n_modes])
            d.n_dof = ms.shape[1]
            d.mode_shapes = _dp(ms)
        self._chk(self._l.pbso_group_add_object(self._h, global_id, C.byref(d)))

    def finalize(self):
        self._chk(self._l.pbso_group_finalize(self._h))

    def engine(self, rank):
        """the Engine wrapper of a LOCAL rank (its object ids are local: global id - span(rank)[0])"""
        if rank not in self._engines:
            h = self._l.pbso_group_engine(self._h, rank)
            if not h:
                raise PbsoError(capi.ERR_INVALID, f"rank {rank} belongs to another process")
            lo, hi = self.span(rank)
            self._engines[rank] = Engine.from_handle(h, [int(x) for x in self._modes[lo:hi]], self.qnorm_mode, self.B)
        return self._engines[rank]

    def enqueue_force(self, global_id, msg, not_before=0):
        return bool(self._chk(self._l.pbso_group_enqueue_force(self._h, global_id, C.byref(msg.to_c()), not_before)))

    # -- stepping -----------------------------------------------------------------
    def step(self, n_buffers):
        self._chk(self._l.pbso_group_step(self._h, n_buffers))
        self._last_nb = n_buffers
        for e in self._engines.values():
            e._last_nb = n_buffers
            e._borrowed = None

    def gather(self, mode=capi.GATHER_ALL):
        self._chk(self._l.pbso_group_gather(self._h, mode))

    def sync(self):
        self._chk(self._l.pbso_group_sync(self._h))

    def result_ptr(self, rank):
        rows, row = C.c_size_t(0), C.c_size_t(0)
        p = self._l.pbso_group_result_device_ptr(self._h, rank, C.byref(rows), C.byref(row))
        return p, rows.value, row.value

    def result(self, rank):
        """the last gather's result on a local rank as a numpy array [rows][n_buffers * 513] (synchronous)"""
        p, rows, row = self.result_ptr(rank)
        if not p:
            raise PbsoError(capi.ERR_STATE, "no gather result on that rank")
        out = np.empty((rows, row), dtype=np.float32)
        self._chk(self._l.pbso_group_read_result(self._h, rank, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out
