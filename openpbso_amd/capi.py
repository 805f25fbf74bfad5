"""ctypes binding of include/openpbso_amd.h (libopenpbso_amd.so).

Fails loudly when the library is missing -- the HIP path is the only path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBSO_LIB") or os.path.join(_HERE, "libopenpbso_amd.so")      # PBSO_LIB: A/B runs of two builds in one process tree

ABI_VERSION = 6
OK = 0
ERR_INVALID, ERR_HIP, ERR_STATE, ERR_IO, ERR_MISSING_MAP, ERR_ASSERT, ERR_NOMEM = -1, -2, -3, -4, -5, -6, -7
POINT_FORCE, GAUSSIAN_FORCE, AUTOREGRESSIVE_FORCE = 0, 1, 2
DATA_EXPLICIT, DATA_VERTEX, DATA_FACE, DATA_ZERO = 0, 1, 2, 3
FORM_BLOCK, FORM_VELOCITY, FORM_DIRECT, FORM_BLOCK_BF16 = 0, 1, 2, 3
QNORM_OFF, QNORM_ALL, QNORM_CLOSED = 0, 1, 2

# every symbol include/openpbso_amd.h declares
EXPORTS = [
    "pbso_abi_version", "pbso_status_string", "pbso_engine_create", "pbso_engine_destroy",
    "pbso_last_error", "pbso_add_object", "pbso_add_object_from_files", "pbso_object_set_ffat_maps",
    "pbso_object_read_ffat_maps", "pbso_fatcube_parse", "pbso_ffat_map_free", "pbso_finalize",
    "pbso_enqueue_force", "pbso_enqueue_force_batch", "pbso_enqueue_vertex_hits", "pbso_enqueue_arprm", "pbso_compute_transfer", "pbso_compute_transfer_batch", "pbso_object_n_maps", "pbso_listeners_enable", "pbso_mix_listeners",
    "pbso_set_use_transfer", "pbso_get_latest_transfer", "pbso_step", "pbso_step_into", "pbso_sync",
    "pbso_read_audio", "pbso_read_emitted", "pbso_read_qnorm", "pbso_read_state", "pbso_write_state",
    "pbso_audio_device_ptr", "pbso_pa_convert", "pbso_get_info",
    "pbso_modes_read", "pbso_num_modes_audible", "pbso_material_read", "pbso_free", "pbso_read_census", "pbso_obj_read", "pbso_arprm_pending", "pbso_flush",
    "pbso_mix_objects", "pbso_read_audio_rows", "pbso_compute_transfer_path",
    "pbso_step_to_host", "pbso_host_wait", "pbso_host_alloc", "pbso_host_free",
    # the device group (one engine per GPU, RCCL gather called from C++)
    "pbso_group_unique_id", "pbso_group_create", "pbso_group_destroy", "pbso_group_last_error", "pbso_group_plan",
    "pbso_group_rank_span", "pbso_group_owner", "pbso_group_add_object", "pbso_group_finalize", "pbso_group_engine",
    "pbso_group_enqueue_force", "pbso_group_step", "pbso_group_gather", "pbso_group_sync", "pbso_group_result_device_ptr",
    "pbso_group_read_result", "pbso_shard_by_modes",
]
GATHER_ALL, GATHER_ROOT, GATHER_MIX = 1, 2, 3
GROUP_ID_BYTES = 128


class GroupDesc(C.Structure):
    pass          # (fields follow EngineDesc: set below)


class EngineDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int), ("device", C.c_int), ("frames_per_buffer", C.c_int),
                ("sample_rate", C.c_int), ("recurrence_form", C.c_int), ("qnorm_mode", C.c_int),
                ("modes_per_lane", C.c_int), ("stream", C.c_void_p),
                # ABI 4: kernel / path selection per engine (0 = the engine's policy)
                ("bank_kernel", C.c_int), ("time_chunks", C.c_int), ("direct_hits", C.c_int), ("forced_block", C.c_int),
                ("dense_launches", C.c_int), ("device_profiles", C.c_int), ("profile_kernel", C.c_int),
                ("profile_margin_pct", C.c_int), ("profile_priority", C.c_int), ("team_waves", C.c_int),
                ("pipe_consumers", C.c_int), ("pipe_max_teams", C.c_longlong), ("chunk_buffers", C.c_int),
                ("plan_threads", C.c_int), ("plan_pin", C.c_int), ("timing_every", C.c_int), ("warm_copies", C.c_int),
                ("stream_sync", C.c_int), ("latency_path", C.c_int), ("time_chunk_shape", C.c_int), ("scan_kernel", C.c_int), ("fuse_short_launches", C.c_int), ("submit_thread", C.c_int)]
BANK_AUTO, BANK_BLOCK, BANK_PIPE = 0, 1, 2


class ObjectDesc(C.Structure):
    _fields_ = [("n_modes", C.c_int), ("n_omega", C.c_int), ("omega_squared", C.POINTER(C.c_double)),
                ("density", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("n_dof", C.c_int), ("mode_shapes", C.POINTER(C.c_double))]


class FfatMap(C.Structure):
    _fields_ = [("mode_id", C.c_int), ("k", C.c_double), ("center3", C.c_double * 3),
                ("cell_size", C.c_double), ("low_corners", (C.c_double * 3) * 6),
                ("n_elements", (C.c_int * 2) * 6), ("strides", C.c_int * 6),
                ("center", C.c_double * 3), ("bbox_low", C.c_double * 3), ("bbox_top", C.c_double * 3),
                ("n_psi", C.c_int), ("psi", C.POINTER(C.c_double))]


class ForceMsg(C.Structure):
    _fields_ = [("force_type", C.c_int), ("gaussian_width_us", C.c_double),
                ("sustained_force_start", C.c_int), ("sustained_force_end", C.c_int),
                ("clear_all_forces", C.c_int), ("data_kind", C.c_int),
                ("data", C.POINTER(C.c_double)), ("n_data", C.c_int), ("vids", C.c_int * 3),
                ("coords", C.c_double * 3), ("vn", C.c_double * 3)]


class EngineInfo(C.Structure):
    _fields_ = [("n_objects", C.c_int), ("frames_per_buffer", C.c_int), ("modes_padded", C.c_int),
                ("modes_per_lane", C.c_int), ("waves_per_object", C.c_int),
                ("lds_bytes_per_workgroup", C.c_int), ("buffers_done", C.c_int64),
                ("last_step_kernel_ms", C.c_double), ("last_step_device_ms", C.c_double),
                ("last_step_host_plan_ms", C.c_double), ("last_step_forced_rows", C.c_int64),
                ("last_step_transfer_rows", C.c_int64),
                ("total_kernel_ms", C.c_double), ("total_device_ms", C.c_double),
                ("total_host_plan_ms", C.c_double), ("total_steps", C.c_int64), ("n_teams", C.c_int),
                ("recurrence_form", C.c_int), ("total_block_launches", C.c_int64), ("total_sample_launches", C.c_int64),
                ("total_timed_launches", C.c_int64), ("total_split_launches", C.c_int64),
                ("total_time_chunk_launches", C.c_int64), ("total_dropped_hits", C.c_int64), ("total_one_stream_launches", C.c_int64),
                ("total_dense_increment_launches", C.c_int64), ("total_segmented_scans", C.c_int64), ("last_time_chunk_shape", C.c_int), ("last_time_chunk_buffers", C.c_int),
                ("last_time_chunk_teams", C.c_int), ("start_gate", C.c_int), ("total_gate_timeouts", C.c_int64),
                ("total_host_submit_ms", C.c_double), ("total_ffat_shared_events", C.c_int64), ("total_ffat_general_events", C.c_int64)]


GroupDesc._fields_ = [("abi_version", C.c_int), ("devices", C.POINTER(C.c_int)), ("n_devices", C.c_int), ("world_size", C.c_int),
                      ("first_rank", C.c_int), ("unique_id", C.c_void_p), ("engine", EngineDesc), ("transport", C.c_int)]
GROUP_RCCL, GROUP_RCCL_ALWAYS, GROUP_LOOPBACK = 0, 1, 2

_lib = None


def lib():
    """Load libopenpbso_amd.so (raises if it is not built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP engine first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    l = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    l.pbso_abi_version.restype = C.c_int
    l.pbso_status_string.restype = C.c_char_p
    l.pbso_status_string.argtypes = [C.c_int]
    l.pbso_engine_create.argtypes = [C.POINTER(EngineDesc), C.POINTER(vp)]
    l.pbso_engine_destroy.argtypes = [vp]
    l.pbso_engine_destroy.restype = None
    l.pbso_last_error.restype = C.c_char_p
    l.pbso_last_error.argtypes = [vp]
    l.pbso_add_object.argtypes = [vp, C.POINTER(ObjectDesc), ip]
    l.pbso_add_object_from_files.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, ip, ip]
    l.pbso_object_set_ffat_maps.argtypes = [vp, C.c_int, C.POINTER(FfatMap), C.c_int]
    l.pbso_object_read_ffat_maps.argtypes = [vp, C.c_int, C.c_char_p]
    l.pbso_fatcube_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(FfatMap)]
    l.pbso_ffat_map_free.argtypes = [C.POINTER(FfatMap)]
    l.pbso_ffat_map_free.restype = None
    l.pbso_finalize.argtypes = [vp]
    l.pbso_enqueue_force.argtypes = [vp, C.c_int, C.POINTER(ForceMsg), C.c_int64]
    l.pbso_enqueue_vertex_hits.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    l.pbso_enqueue_force_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(ForceMsg), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_ubyte)]
    l.pbso_enqueue_arprm.argtypes = [vp, C.c_int, dp, C.c_double, C.c_double, C.c_int64]
    l.pbso_arprm_pending.argtypes = [vp, C.c_int]
    l.pbso_flush.argtypes = [vp]
    l.pbso_compute_transfer.argtypes = [vp, C.c_int, dp, C.c_int64]
    l.pbso_compute_transfer_batch.argtypes = [vp, C.c_int, dp, C.c_int, dp, C.c_int]
    l.pbso_object_n_maps.argtypes = [vp, C.c_int]
    l.pbso_listeners_enable.argtypes = [vp, C.c_int]
    l.pbso_mix_listeners.argtypes = [vp, C.c_int, dp, C.c_int, C.POINTER(C.c_float), C.c_size_t]
    l.pbso_set_use_transfer.argtypes = [vp, C.c_int, C.c_int, C.c_int64]
    l.pbso_get_latest_transfer.argtypes = [vp, C.c_int, dp]
    l.pbso_step.argtypes = [vp, C.c_int]
    l.pbso_step_into.argtypes = [vp, C.c_int, vp]
    l.pbso_sync.argtypes = [vp]
    l.pbso_read_audio.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t]
    l.pbso_read_emitted.argtypes = [vp, C.POINTER(C.c_ubyte), C.c_size_t]
    l.pbso_read_qnorm.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]
    l.pbso_read_state.argtypes = [vp, C.c_int, dp, dp, C.c_int]
    l.pbso_write_state.argtypes = [vp, C.c_int, dp, dp, C.c_int]
    l.pbso_audio_device_ptr.restype = vp
    l.pbso_audio_device_ptr.argtypes = [vp]
    l.pbso_pa_convert.argtypes = [C.POINTER(C.c_float), C.c_ulong, C.POINTER(C.c_float)]
    l.pbso_pa_convert.restype = None
    l.pbso_get_info.argtypes = [vp, C.POINTER(EngineInfo)]
    l.pbso_modes_read.argtypes = [C.c_char_p, ip, ip, C.POINTER(dp), C.POINTER(dp)]
    l.pbso_num_modes_audible.argtypes = [dp, C.c_int, C.c_double, C.c_double]
    l.pbso_material_read.argtypes = [C.c_char_p, dp]
    l.pbso_obj_read.argtypes = [C.c_char_p, ip, ip, C.POINTER(dp), C.POINTER(ip), C.POINTER(dp)]
    l.pbso_read_census.argtypes = [vp, C.POINTER(C.c_uint64), C.c_size_t]
    l.pbso_free.argtypes = [vp]
    l.pbso_free.restype = None
    l.pbso_mix_objects.argtypes = [vp, vp]
    l.pbso_step_to_host.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_size_t]
    l.pbso_host_wait.argtypes = [vp]
    l.pbso_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    l.pbso_host_free.argtypes = [vp]
    l.pbso_host_free.restype = None
    l.pbso_compute_transfer_path.argtypes = [vp, C.c_int, ip, dp, C.POINTER(C.c_int64), C.POINTER(C.c_ubyte)]
    l.pbso_read_audio_rows.argtypes = [vp, ip, C.c_int, C.POINTER(C.c_float)]
    l.pbso_shard_by_modes.argtypes = [ip, C.c_int, C.c_int, ip]
    l.pbso_group_unique_id.argtypes = [vp]
    l.pbso_group_create.argtypes = [C.POINTER(GroupDesc), C.POINTER(vp)]
    l.pbso_group_destroy.argtypes = [vp]
    l.pbso_group_destroy.restype = None
    l.pbso_group_last_error.argtypes = [vp]
    l.pbso_group_last_error.restype = C.c_char_p
    l.pbso_group_plan.argtypes = [vp, ip, C.c_int]
    l.pbso_group_rank_span.argtypes = [vp, C.c_int, ip, ip]
    l.pbso_group_owner.argtypes = [vp, C.c_int, ip, ip]
    l.pbso_group_add_object.argtypes = [vp, C.c_int, C.POINTER(ObjectDesc)]
    l.pbso_group_finalize.argtypes = [vp]
    l.pbso_group_engine.argtypes = [vp, C.c_int]
    l.pbso_group_engine.restype = vp
    l.pbso_group_enqueue_force.argtypes = [vp, C.c_int, C.POINTER(ForceMsg), C.c_int64]
    l.pbso_group_step.argtypes = [vp, C.c_int]
    l.pbso_group_gather.argtypes = [vp, C.c_int]
    l.pbso_group_sync.argtypes = [vp]
    l.pbso_group_result_device_ptr.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    l.pbso_group_result_device_ptr.restype = vp
    l.pbso_group_read_result.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_size_t]
    _lib = l
    return l
