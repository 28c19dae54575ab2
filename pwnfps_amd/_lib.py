"""ctypes binding of libpwnhip.so (include/pwnhip.h).

The library is the product: there is no CPU fallback.  If it is missing or
does not export the ABI, importing this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PWNHIP_LIB selects an experiment build (make VARIANT=tag) of the same ABI
LIB_PATH = os.environ.get("PWNHIP_LIB") or os.path.join(_HERE, "libpwnhip.so")

PWN_OK, PWN_EINVAL, PWN_ENODEV, PWN_ENOMEM, PWN_EIO, PWN_EHIP, PWN_ENOLEVEL, PWN_ETOOBIG = 0, -1, -2, -3, -4, -5, -6, -7
PWN_EBUSY, PWN_ENOTSUP, PWN_ETIMEDOUT = -8, -9, -10
PWN_OPT_BLUR_PASSES, PWN_OPT_COUNTERS, PWN_OPT_SCHEDULER, PWN_OPT_REFILL_LIMIT, PWN_OPT_FRAME_TIMING, PWN_OPT_WAVE_LOG = 1, 2, 3, 4, 5, 6
PWN_OPT_FRAME_OVERLAP = 7
PWN_OPT_TRACE_ROOM = 8
PWN_OPT_UNIT_ORDER = 9
PWN_OPT_TILED_CHOREO = 10
PWN_OPT_TILED_STREAMS = 11
PWN_OPT_TILED_COMMS = 12
PWN_OPT_CALL_STRIPS = 13
PWN_TILED_CHOREO_INSTREAM, PWN_TILED_CHOREO_SPLIT = 0, 1
PWN_SCHED_UNITS, PWN_SCHED_REFILL = 0, 1
PWN_MAX_SLOTS = 4
PWN_FRAME_SBUF, PWN_FRAME_ZBUF, PWN_FRAME_SURFACE = 1, 2, 4
PWN_OBJ_MAX = 10000
(PROBE_RCP, PROBE_RSQRT, PROBE_SINF, PROBE_COSF, PROBE_EXPF, PROBE_SQRT, PROBE_DIV,
 PROBE_FTOINT, PROBE_RANDFS, PROBE_SIN_OF_PAIR, PROBE_COS_OF_PAIR) = range(11)


class Portal(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("x1", "z1", "x2", "z2", "rot12", "c1", "c2")]


class Sphere(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("r", "refl", "x", "y", "z", "cb", "cg", "cr")]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("steps", C.c_uint64), ("portals", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("exhausted", C.c_uint64), ("wave_steps", C.c_uint64),
                ("trace_ms", C.c_float), ("blur_ms", C.c_float), ("total_ms", C.c_float), ("reserved_", C.c_float),
                ("wave_paths", C.c_uint64 * 8), ("phase_passes", C.c_uint64), ("phase_lanes", C.c_uint64),
                ("wave_time", C.c_uint64), ("kernel_span", C.c_uint64), ("waves", C.c_uint64), ("regions", C.c_uint64 * 32)]


class Frame(C.Structure):
    _fields_ = [("sbuf", C.c_void_p), ("zbuf", C.c_void_p), ("surface", C.c_void_p), ("surface_pitch_bytes", C.c_int),
                ("d_sbuf", C.c_void_p), ("d_zbuf", C.c_void_p), ("d_surface", C.c_void_p),
                ("sec_current", C.c_float), ("trace_ms", C.c_float), ("blur_ms", C.c_float), ("sink_ms", C.c_float), ("timed", C.c_int),
                ("seq", C.c_uint64)]


class TiledFrame(C.Structure):
    _fields_ = [("d_sbuf", C.c_void_p), ("sbuf", C.c_void_p), ("seq", C.c_uint64), ("redone", C.c_int),
                ("timed", C.c_int), ("trace_ms", C.c_float), ("frame_ms", C.c_float), ("blur_ms", C.c_float),
                ("halo_ms", C.c_float), ("gather_ms", C.c_float), ("enqueue_us", C.c_float),
                ("y0", C.c_int), ("y1", C.c_int), ("cost", C.c_uint32), ("root", C.c_int)]


class TiledInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("rank", "world", "y0", "y1", "rows_per_rank", "halo_rows", "transport")] + \
               [(n, C.c_uint64) for n in ("frames", "frames_redone", "groups", "bytes_sent", "bytes_received", "bytes_to_host")] + \
               [(n, C.c_int) for n in ("host_sink", "max_rows", "balance_every", "grid_reserve", "two_streams")] + \
               [("recuts", C.c_uint64), ("gather_root", C.c_int), ("rccl_nonblocking", C.c_int),
                ("init_timeout_ms", C.c_int), ("wait_timeout_ms", C.c_int), ("dead", C.c_int), ("compute_streams", C.c_int), ("choreography", C.c_int), ("communicators", C.c_int)]


class GroupInfo(C.Structure):
    _fields_ = [("members", C.c_int), ("transport", C.c_int), ("devices", C.c_int * 64), ("cuts", C.c_int * 65),
                ("halo_rows", C.c_int), ("host_sink", C.c_int), ("frames", C.c_uint64), ("frames_redone", C.c_uint64), ("recuts", C.c_uint64), ("note", C.c_char * 160)]


PWN_TILED_ID_BYTES = 128
PWN_TRANSPORT_RCCL, PWN_TRANSPORT_SHM, PWN_TRANSPORT_LOCAL = 0, 1, 2
PWN_TILED_HOST = 1
PWN_TILED_SLOTS = 6
PWN_TILED_MAX_WORLD = 64

# every symbol include/pwnhip.h declares: (name, restype, argtypes)
_vp, _i, _f, _d = C.c_void_p, C.c_int, C.c_float, C.c_double
ABI = [
    ("pwn_init", _i, [C.POINTER(_vp), _i, _i, _i]),
    ("pwn_init_multi", _i, [C.POINTER(_vp), _vp, _i, _i, _i]),
    ("pwn_group_info_get", _i, [_vp, C.POINTER(GroupInfo)]),
    ("pwn_destroy", None, [_vp]),
    ("pwn_set_option", _i, [_vp, _i, _i]),
    ("pwn_trace_room_state", _i, [_vp, _vp]),
    ("pwn_strerror", C.c_char_p, [_i]),
    ("pwn_last_error", C.c_char_p, [_vp]),
    ("pwn_level_load", _i, [_vp, C.c_char_p]),
    ("pwn_level_load_mem", _i, [_vp, C.c_char_p, _i]),
    ("pwn_upload_level", _i, [_vp, _vp, _vp]),
    ("pwn_get_level", _i, [_vp, _vp, _vp, _vp]),
    ("pwn_upload_spheres", _i, [_vp, _vp, _i]),
    ("pwn_get_bins", _i, [_vp, _vp, _vp, _i]),
    ("pwn_obj_new", _i, [_vp]),
    ("pwn_obj_set_sphere", _i, [_vp, _i, _d, _d, _d, _d, _d, _d, _d, _d]),
    ("pwn_obj_free", _i, [_vp, _i]),
    ("pwn_level_get", _i, [_vp, _i, _i]),
    ("pwn_prepare_render", _i, [_vp]),
    ("pwn_get_objects", _i, [_vp, _vp, _i]),
    ("pwn_trace_screen_centred", _i, [_vp, _vp, _f, _vp, _vp]),
    ("pwn_host_register", _i, [_vp, _vp, C.c_size_t]),
    ("pwn_host_unregister", _i, [_vp, _vp]),
    ("pwn_call_strips_state", _i, [_vp, _vp]),
    ("pwn_frames_config", _i, [_vp, _i, _i, _i, _i]),
    ("pwn_submit_frame", _i, [_vp, _vp, _f, _i]),
    ("pwn_wait_frame", _i, [_vp, _i, C.POINTER(Frame)]),
    ("pwn_frame_ready", _i, [_vp, _i]),
    ("pwn_read_plane", _i, [_vp, _vp, _vp, C.c_size_t]),
    ("pwn_trace_rows_device", _i, [_vp, _vp, _f, _i, _i, _vp, _vp, _vp]),
    ("pwn_blur_rows_device", _i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    ("pwn_blur_rows_device_bounded", _i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    ("pwn_unit_order_state", _i, [_vp, _vp]),
    ("pwn_launch_order_waits", _i, [_vp, _vp]),
    ("pwn_unit_order_probe", _i, [_vp, _vp, C.c_uint32, _vp]),
    ("pwn_tiled_unique_id", _i, [_vp, _i]),
    ("pwn_tiled_init", _i, [_vp, _i, _i, _vp, _i, _i]),
    ("pwn_tiled_submit", _i, [_vp, _vp, _f]),
    ("pwn_tiled_wait", _i, [_vp, _i, C.POINTER(TiledFrame)]),
    ("pwn_tiled_host_sink", _i, [_vp, _vp, C.c_size_t]),
    ("pwn_tiled_gather_root", _i, [_vp, _i]),
    ("pwn_tiled_get_info", _i, [_vp, C.POINTER(TiledInfo)]),
    ("pwn_tiled_shutdown", None, [_vp]),
    ("pwn_tiled_set_timeouts", _i, [_vp, _i, _i]),
    ("pwn_tiled_preflight", _i, [_vp, C.c_char_p, C.c_size_t]),
    ("pwn_tiled_balance", _i, [_vp, _i]),
    ("pwn_tiled_set_cuts", _i, [_vp, _vp, _i]),
    ("pwn_tiled_get_cuts", _i, [_vp, _vp, _vp]),
    ("pwn_tiled_set_reserve", _i, [_vp, _i]),
    ("pwn_tiled_recut", _i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    ("pwn_screen_upscale", _i, [_vp, _vp, _i, _i, _vp]),
    ("pwn_upscale_device", _i, [_vp, _vp, _i, _i, _vp, _vp]),
    ("pwn_get_stats", _i, [_vp, C.POINTER(Stats)]),
    ("pwn_probe", _i, [_vp, _i, _vp, _vp, _i]),
]


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pwnfps_amd: %s is missing -- build it with `make -C pwnfps_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback." % LIB_PATH)
    # One process, one HIP runtime.  The PyTorch wheel ships its own libamdhip64 /
    # libhsa-runtime64; if libpwnhip.so pulled in /opt/rocm's copy first and torch then
    # brought its own, the second runtime to initialise finds no device (pwn_init ->
    # PWN_ENODEV).  Loading torch first makes libpwnhip.so bind to the copy torch uses,
    # which is also what lets the strip entry points take torch's device pointers and
    # streams.  Without torch (the C host) the library uses /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in ABI:
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()
