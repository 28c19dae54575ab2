"""Host-side mirror of the reference's render interface, over libpwnhip.so.

Names follow the reference: ``level_load`` (level.h:107), ``level_prepare_render``
(level.h:64; here ``set_objects``), ``trace_screen_centred`` (screen.h:31),
``screen_upscale`` (screen.h:126).  Errors the reference reports by returning
NULL / asserting are raised as ``PwnError`` with the library's code.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib

SPHERE_DTYPE = np.dtype([("r", "<f4"), ("refl", "<f4"), ("x", "<f4"), ("y", "<f4"),
                         ("z", "<f4"), ("cb", "<f4"), ("cg", "<f4"), ("cr", "<f4")])
PORTAL_DTYPE = np.dtype([(n, "<i4") for n in ("x1", "z1", "x2", "z2", "rot12", "c1", "c2")])


class PwnError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        msg = "%s: %s (%d)" % (where, lib.pwn_strerror(code).decode(), code)
        if detail:
            msg += ": " + detail
        super().__init__(msg)


class Renderer:
    """One GPU context for a fixed frame size (rwidth x rheight, main.c:26-27)."""

    def __init__(self, width, height, device=0, devices=None):
        """devices: a list of HIP ordinals -> pwn_init_multi, ONE handle whose frames are row-tiled over them inside this
        process (the same ordinal several times: so many members on that device)"""
        self.w, self.h, self.device = int(width), int(height), int(device)
        self._ctx = C.c_void_p()
        self.devices = None
        if devices is not None:
            self.devices = [int(d) for d in devices]
            self.device = self.devices[0]
            arr = (C.c_int * len(self.devices))(*self.devices)
            rc = lib.pwn_init_multi(C.byref(self._ctx), arr, len(self.devices), self.w, self.h)
            if rc != 0:
                self._ctx = C.c_void_p()
                raise PwnError(rc, "pwn_init_multi(devices=%s, %dx%d)" % (self.devices, width, height))
            return
        rc = lib.pwn_init(C.byref(self._ctx), self.device, self.w, self.h)
        if rc != 0:
            self._ctx = C.c_void_p()
            raise PwnError(rc, "pwn_init(device=%d, %dx%d)" % (device, width, height))

    def group_info(self):
        gi = _lib.GroupInfo()
        self._chk(lib.pwn_group_info_get(self._ctx, C.byref(gi)), "pwn_group_info_get")
        n = gi.members
        return {"members": n, "transport": {0: "rccl", 1: "shm", 2: "local"}.get(gi.transport, gi.transport), "devices": [gi.devices[i] for i in range(n)],
                "cuts": [gi.cuts[i] for i in range(n + 1)], "halo_rows": gi.halo_rows, "host_sink": bool(gi.host_sink),
                "frames": int(gi.frames), "frames_redone": int(gi.frames_redone), "recuts": int(gi.recuts),
                "note": gi.note.decode("latin-1")}

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            lib.pwn_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc < 0:
            raise PwnError(rc, where, lib.pwn_last_error(self._ctx).decode(errors="replace"))
        return rc

    # -- options -----------------------------------------------------------
    def set_blur_passes(self, n):
        """POSTPROC_BLUR (defs.h:9); 0 disables the post-process."""
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_BLUR_PASSES, int(n)), "pwn_set_option")

    def set_scheduler(self, which):
        """PWN_OPT_SCHEDULER: "units" (a wave64 traces 16x4-pixel units in step) or "refill"
        (lanes whose ray ended are refilled by ballot + prefix rank)."""
        v = {"units": _lib.PWN_SCHED_UNITS, "refill": _lib.PWN_SCHED_REFILL}.get(which, which)
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_SCHEDULER, int(v)), "pwn_set_option")

    def set_frame_timing(self, every):
        """HIP events around the kernels of every N-th frame in flight (0: never, 1: all)."""
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_FRAME_TIMING, int(every)), "pwn_set_option")

    def set_frame_overlap(self, on):
        """frames in flight: successive frames alternate between two compute streams (default) or all run on one"""
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_FRAME_OVERLAP, 1 if on else 0), "pwn_set_option")

    def set_tiled_choreo(self, split):
        """PWN_OPT_TILED_CHOREO, before tiled_init: False (default) = everything of a frame in order on the frame's own compute stream;
        True = the exchanges on a third stream, blur and gather one and two submits late (rounds 2-3).  Same frames."""
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_TILED_CHOREO, 1 if split else 0), "pwn_set_option")

    def set_tiled_comms(self, per_stream):
        """PWN_OPT_TILED_COMMS, before tiled_init: False (default) = one RCCL communicator; True = one per compute stream"""
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_TILED_COMMS, 2 if per_stream else 1), "pwn_set_option")

    def set_tiled_streams(self, n):
        """PWN_OPT_TILED_STREAMS, before tiled_init: compute streams the frames of an in-stream tiling rotate over, 2 or 3"""
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_TILED_STREAMS, int(n)), "pwn_set_option")

    def set_unit_order(self, on):
        """PWN_OPT_UNIT_ORDER: the trace kernel's units handed out by what they cost in the last launch (True) or
        in arithmetic order (False, the default).  Never changes a frame."""
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_UNIT_ORDER, 1 if on else 0), "pwn_set_option")

    def launch_order_waits(self):
        """trace launches so far that left the rotation over the compute streams and were ordered behind the launch R before them"""
        out = C.c_uint64(0)
        self._chk(lib.pwn_launch_order_waits(self._ctx, C.byref(out)), "pwn_launch_order_waits")
        return int(out.value)

    def unit_order_state(self):
        out = (C.c_uint64 * 4)()
        self._chk(lib.pwn_unit_order_state(self._ctx, out), "pwn_unit_order_state")
        return {"option": int(out[0]), "launches_in_sorted_order": int(out[1]), "sorts": int(out[2]), "units_ordered": int(out[3])}

    def unit_order_probe(self, cost):
        """pwn_unit_order_probe: the per-queue sort of `cost` (uint16 per unit) -> [64, cap] uint32, unused entries 0xffffffff"""
        cost = np.ascontiguousarray(cost, np.uint16)
        cap = (cost.size + 63) // 64
        out = np.zeros((64, cap), np.uint32)
        self._chk(lib.pwn_unit_order_probe(self._ctx, cost.ctypes.data, cost.size, out.ctypes.data), "pwn_unit_order_probe")
        return out

    def set_trace_room(self, workgroups):
        """PWN_OPT_TRACE_ROOM: workgroups the persistent trace grid leaves free for the other stream's kernels while frames
        alternate between two streams; -1 (default) = the library measures which of 0 / one per CU is faster and keeps it"""
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_TRACE_ROOM, int(workgroups)), "pwn_set_option")

    def trace_room_state(self):
        out = (C.c_int * 4)()
        self._chk(lib.pwn_trace_room_state(self._ctx, out), "pwn_trace_room_state")
        return {"option": out[0], "room_now": out[1], "comparisons": out[2], "changes": out[3]}

    def set_wave_log(self, on):
        """stats()["wave_time"] / (["waves"] * ["kernel_span"]) = mean wave residency of the last frame"""
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_WAVE_LOG, 1 if on else 0), "pwn_set_option")

    def set_refill_limit(self, n):
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_REFILL_LIMIT, int(n)), "pwn_set_option")

    def set_counters(self, on):
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_COUNTERS, 1 if on else 0), "pwn_set_option")

    # -- level (level.h:107-228) ---------------------------------------------
    def level_load(self, path):
        self._chk(lib.pwn_level_load(self._ctx, str(path).encode()), "pwn_level_load(%s)" % path)

    def level_load_text(self, text):
        if isinstance(text, str):
            text = text.encode("latin-1")
        self._chk(lib.pwn_level_load_mem(self._ctx, text, len(text)), "pwn_level_load_mem")

    def upload_level(self, data, pmap):
        data = np.ascontiguousarray(data, np.uint8)
        pmap = np.ascontiguousarray(pmap, np.int32)
        if data.shape != (64, 64) or pmap.shape != (26, 7):
            raise ValueError("data must be (64,64) uint8 and pmap (26,7) int32")
        self._chk(lib.pwn_upload_level(self._ctx, data.ctypes.data, pmap.ctypes.data), "pwn_upload_level")

    def get_level(self):
        data = np.zeros((64, 64), np.uint8)
        pmap = np.zeros((26, 7), np.int32)
        spawn = np.zeros(2, np.int32)
        self._chk(lib.pwn_get_level(self._ctx, data.ctypes.data, pmap.ctypes.data, spawn.ctypes.data), "pwn_get_level")
        return data, pmap, spawn

    # -- objects (script.h:10-40 + level.h:64-81) ----------------------------
    def set_objects(self, spheres):
        """The live sphere set for the next frames; binning per cell
        (level_prepare_render) happens inside."""
        spheres = np.ascontiguousarray(spheres, SPHERE_DTYPE)
        self._chk(lib.pwn_upload_spheres(self._ctx, spheres.ctypes.data if len(spheres) else None,
                                         len(spheres)), "pwn_upload_spheres")

    def obj_new(self):
        """obj_new() (script.h:1-8): index of a fresh object slot."""
        return self._chk(lib.pwn_obj_new(self._ctx), "pwn_obj_new")

    def obj_set(self, obj, typ, r, refl, x, y, z, cb, cg, cr):
        """obj_set(o, "sphere", r, refl, x, y, z, b, g, r) (script.h:10-40)."""
        if str(typ).lower() != "sphere":
            raise ValueError('obj_set: invalid typ "%s"' % typ)
        self._chk(lib.pwn_obj_set_sphere(self._ctx, int(obj), r, refl, x, y, z, cb, cg, cr), "pwn_obj_set_sphere")
        return obj

    def obj_free(self, obj):
        self._chk(lib.pwn_obj_free(self._ctx, int(obj)), "pwn_obj_free")

    def level_get(self, cx, cz):
        """level_get(cx, cz) (script.h:53-64): the cell as a 1-character string."""
        return chr(self._chk(lib.pwn_level_get(self._ctx, int(cx), int(cz)), "pwn_level_get"))

    def level_prepare_render(self):
        """level_prepare_render (level.h:64-81) over the object table."""
        self._chk(lib.pwn_prepare_render(self._ctx), "pwn_prepare_render")

    def get_objects(self):
        n = self._chk(lib.pwn_get_objects(self._ctx, None, 0), "pwn_get_objects")
        out = np.zeros(n, SPHERE_DTYPE)
        if n:
            self._chk(lib.pwn_get_objects(self._ctx, out.ctypes.data, n), "pwn_get_objects")
        return out

    def get_bins(self):
        counts = np.zeros(4096, np.uint16)
        n = self._chk(lib.pwn_get_bins(self._ctx, counts.ctypes.data, None, 0), "pwn_get_bins")
        idx = np.zeros(max(n, 1), np.int32)
        self._chk(lib.pwn_get_bins(self._ctx, counts.ctypes.data, idx.ctypes.data, n), "pwn_get_bins")
        return counts, idx[:n]

    # -- frame (screen.h:31-124) ---------------------------------------------
    def trace_screen_centred(self, cam, sec_current=0.0, want_z=True, sbuf=None, zbuf=None):
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        if sbuf is None:
            sbuf = np.empty((self.h, self.w), np.uint32)
        if want_z and zbuf is None:
            zbuf = np.empty((self.h, self.w), np.float32)
        self._chk(lib.pwn_trace_screen_centred(self._ctx, cam.ctypes.data, float(sec_current),
                                               sbuf.ctypes.data, zbuf.ctypes.data if want_z else None),
                  "pwn_trace_screen_centred")
        return (sbuf, zbuf) if want_z else sbuf

    def set_call_strips(self, n):
        """PWN_OPT_CALL_STRIPS: -1 = by frame size (default), 0 = one launch per pass, 2..32 = that many row strips"""
        self._chk(lib.pwn_set_option(self._ctx, _lib.PWN_OPT_CALL_STRIPS, int(n)), "pwn_set_option(CALL_STRIPS)")

    def call_strips_state(self):
        v = (C.c_ulonglong * 6)()
        self._chk(lib.pwn_call_strips_state(self._ctx, v), "pwn_call_strips_state")
        opt = int(v[0])
        return {"option": opt - (1 << 64) if opt >= (1 << 63) else opt, "strips_last": int(v[1]), "calls_in_strips": int(v[2]), "redone": int(v[3]),
                "copy_streams": int(v[4]), "reach_depth": int(v[5])}

    def host_register(self, arr):
        """pwn_host_register: the host's frame buffer (main.c:395-400), made known to the device once"""
        self._chk(lib.pwn_host_register(self._ctx, arr.ctypes.data, arr.nbytes), "pwn_host_register")

    def host_unregister(self, arr):
        self._chk(lib.pwn_host_unregister(self._ctx, arr.ctypes.data), "pwn_host_unregister")

    # -- frames in flight (main.c:93-109 with the hand-over to the host overlapped) ----------
    def frames_config(self, nslots, sbuf=True, zbuf=False, surface_scale=0, pitch_bytes=0):
        flags = (_lib.PWN_FRAME_SBUF if sbuf else 0) | (_lib.PWN_FRAME_ZBUF if zbuf else 0) | \
                (_lib.PWN_FRAME_SURFACE if surface_scale else 0)
        self._chk(lib.pwn_frames_config(self._ctx, int(nslots), flags, int(surface_scale), int(pitch_bytes)), "pwn_frames_config")
        self._frame_scale = int(surface_scale)

    def submit_frame(self, cam, sec_current, slot):
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        self._chk(lib.pwn_submit_frame(self._ctx, cam.ctypes.data, float(sec_current), int(slot)), "pwn_submit_frame")

    def frame_ready(self, slot):
        return bool(self._chk(lib.pwn_frame_ready(self._ctx, int(slot)), "pwn_frame_ready"))

    def wait_frame(self, slot):
        """Blocks until the slot's frame is on the host.  Returns a dict of numpy VIEWS of the
        library's pinned buffers (valid until the next submit on that slot) and device times."""
        fr = _lib.Frame()
        self._chk(lib.pwn_wait_frame(self._ctx, int(slot), C.byref(fr)), "pwn_wait_frame")
        out = {"seq": fr.seq, "sec": fr.sec_current, "timed": bool(fr.timed),
               "trace_ms": fr.trace_ms, "blur_ms": fr.blur_ms, "sink_ms": fr.sink_ms,
               "d_sbuf": fr.d_sbuf, "d_zbuf": fr.d_zbuf, "d_surface": fr.d_surface}
        n = self.w * self.h

        def view(ptr, ctype, count, shape):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).reshape(shape)
        if fr.sbuf:
            out["sbuf"] = view(fr.sbuf, C.c_uint32, n, (self.h, self.w))
        if fr.zbuf:
            out["zbuf"] = view(fr.zbuf, C.c_float, n, (self.h, self.w))
        if fr.surface:
            pw = fr.surface_pitch_bytes // 4
            out["surface"] = view(fr.surface, C.c_uint32, pw * self.h * self._frame_scale, (self.h * self._frame_scale, pw))
        return out

    def read_plane(self, d_ptr, dtype=np.uint32):
        """Host copy of a (h, w) device plane of a waited-for frame (wait_frame()["d_sbuf"] ...)."""
        out = np.empty((self.h, self.w), dtype)
        self._chk(lib.pwn_read_plane(self._ctx, C.c_void_p(d_ptr), out.ctypes.data, out.nbytes), "pwn_read_plane")
        return out

    def trace_rows_device(self, cam, sec_current, y0, y1, d_sbuf, d_zbuf, stream=0):
        """Rows [y0,y1) into device frames given as raw device pointers."""
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        self._chk(lib.pwn_trace_rows_device(self._ctx, cam.ctypes.data, float(sec_current), int(y0), int(y1),
                                            C.c_void_p(d_sbuf), C.c_void_p(d_zbuf), C.c_void_p(stream)),
                  "pwn_trace_rows_device")

    def blur_rows_device(self, y0, y1, d_pre, d_zbuf, d_out, stream=0):
        self._chk(lib.pwn_blur_rows_device(self._ctx, int(y0), int(y1), C.c_void_p(d_pre), C.c_void_p(d_zbuf),
                                           C.c_void_p(d_out), C.c_void_p(stream)), "pwn_blur_rows_device")

    def blur_rows_device_bounded(self, y0, y1, d_pre, d_zbuf, d_out, avail_y0, avail_y1, d_miss, stream=0):
        """blur_rows_device when only rows [avail_y0, avail_y1) of d_pre are this frame's;
        taps outside add to the uint32 at d_miss."""
        self._chk(lib.pwn_blur_rows_device_bounded(self._ctx, int(y0), int(y1), C.c_void_p(d_pre), C.c_void_p(d_zbuf),
                                                   C.c_void_p(d_out), int(avail_y0), int(avail_y1), C.c_void_p(d_miss),
                                                   C.c_void_p(stream)), "pwn_blur_rows_device_bounded")

    # -- row tiling over the GPUs of a node (one process per GPU; RCCL inside the library) ----
    @staticmethod
    def tiled_unique_id(transport="rccl"):
        """rank 0: 128 bytes to hand to the other ranks"""
        buf = C.create_string_buffer(_lib.PWN_TILED_ID_BYTES)
        rc = lib.pwn_tiled_unique_id(buf, _lib.PWN_TRANSPORT_SHM if transport == "shm" else _lib.PWN_TRANSPORT_RCCL)
        if rc != 0:
            raise PwnError(rc, "pwn_tiled_unique_id(%s)" % transport)
        return buf.raw

    def tiled_init(self, rank, world, uid, transport="rccl", halo_rows=-1):
        self._chk(lib.pwn_tiled_init(self._ctx, int(rank), int(world), C.c_char_p(uid),
                                     _lib.PWN_TRANSPORT_SHM if transport == "shm" else _lib.PWN_TRANSPORT_RCCL, int(halo_rows)),
                  "pwn_tiled_init(rank %d of %d, %s)" % (rank, world, transport))

    def tiled_submit(self, cam, sec_current=0.0):
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        self._chk(lib.pwn_tiled_submit(self._ctx, cam.ctypes.data, float(sec_current)), "pwn_tiled_submit")

    def tiled_wait(self, host=False):
        """Oldest frame in flight.  The frame's root (rank 0, or rank (seq - 1) mod world after tiled_gather_root(True)):
        dict with d_sbuf (device pointer) and, with host=True, sbuf (numpy view of the pinned copy); other ranks: dict
        without them ("root" says which rank has it).  With a host sink
        (tiled_host_sink) every rank gets sbuf, a view of the frame in the shared host memory."""
        fr = _lib.TiledFrame()
        self._chk(lib.pwn_tiled_wait(self._ctx, _lib.PWN_TILED_HOST if host else 0, C.byref(fr)), "pwn_tiled_wait")
        out = {"seq": fr.seq, "redone": bool(fr.redone), "d_sbuf": fr.d_sbuf, "timed": bool(fr.timed),
               "trace_ms": fr.trace_ms, "frame_ms": fr.frame_ms, "blur_ms": fr.blur_ms, "halo_ms": fr.halo_ms,
               "gather_ms": fr.gather_ms, "enqueue_us": fr.enqueue_us, "y0": fr.y0, "y1": fr.y1, "cost": fr.cost, "root": fr.root}
        if fr.sbuf:
            out["sbuf"] = np.ctypeslib.as_array(C.cast(fr.sbuf, C.POINTER(C.c_uint32)), shape=(self.w * self.h,)).reshape(self.h, self.w)
        return out

    def tiled_host_sink(self, buf):
        """Deliver frames to the host from every rank: `buf` is writable host memory for PWN_TILED_SLOTS whole
        frames (a numpy array, an mmap of POSIX shared memory ...), the same memory in every rank."""
        arr = np.frombuffer(buf, np.uint8)
        self._host_sink = (buf, arr)                      # keep it mapped for as long as the context lives
        self._chk(lib.pwn_tiled_host_sink(self._ctx, arr.ctypes.data, arr.size), "pwn_tiled_host_sink")

    def tiled_gather_root(self, rotate):
        """pwn_tiled_gather_root: frames gathered on rank 0 (False, the default) or on rank f mod world in turn (True);
        the same call on every rank, with no frame in flight."""
        self._chk(lib.pwn_tiled_gather_root(self._ctx, 1 if rotate else 0), "pwn_tiled_gather_root")

    def tiled_info(self):
        inf = _lib.TiledInfo()
        self._chk(lib.pwn_tiled_get_info(self._ctx, C.byref(inf)), "pwn_tiled_get_info")
        return {n: getattr(inf, n) for n, _ in _lib.TiledInfo._fields_}

    def tiled_shutdown(self):
        lib.pwn_tiled_shutdown(self._ctx)

    def tiled_set_timeouts(self, init_s=None, wait_s=None):
        """pwn_tiled_set_timeouts: how long pwn_tiled_init / pwn_tiled_wait may wait for the other ranks before they
        return PWN_ETIMEDOUT (seconds; None keeps a value, a negative number restores the default)."""
        ms = [0 if v is None else (-1 if v < 0 else max(1, int(v * 1000))) for v in (init_s, wait_s)]
        self._chk(lib.pwn_tiled_set_timeouts(self._ctx, ms[0], ms[1]), "pwn_tiled_set_timeouts")

    def tiled_preflight(self):
        """pwn_tiled_preflight as a dict: visible devices, peer access from this context's device, the librccl that
        dlopen resolved and its version, how the communicator will be driven, the deadlines."""
        import json
        buf = C.create_string_buffer(2048)
        self._chk(lib.pwn_tiled_preflight(self._ctx, buf, len(buf)), "pwn_tiled_preflight")
        return json.loads(buf.value.decode())

    def tiled_balance(self, every_frames):
        """Moving cuts: re-cut the strips every `every_frames` delivered frames from what they cost (0: leave them)."""
        self._chk(lib.pwn_tiled_balance(self._ctx, int(every_frames)), "pwn_tiled_balance")

    def tiled_set_cuts(self, cuts):
        """world + 1 row boundaries for the next submitted frames; the same call on every rank."""
        a = np.ascontiguousarray(cuts, np.int32)
        self._chk(lib.pwn_tiled_set_cuts(self._ctx, a.ctypes.data, len(a)), "pwn_tiled_set_cuts")

    def tiled_get_cuts(self):
        """(cuts of the next frame [world + 1], every rank's cost word of the last delivered frame [world])"""
        cuts = np.zeros(_lib.PWN_TILED_MAX_WORLD + 1, np.int32)
        cost = np.zeros(_lib.PWN_TILED_MAX_WORLD, np.uint32)
        n = self._chk(lib.pwn_tiled_get_cuts(self._ctx, cuts.ctypes.data, cost.ctypes.data), "pwn_tiled_get_cuts")
        return cuts[:n].copy(), cost[:n - 1].copy()

    def tiled_set_reserve(self, workgroups):
        """workgroups the persistent trace grid leaves free for RCCL's kernels (this rank, between frames)"""
        self._chk(lib.pwn_tiled_set_reserve(self._ctx, int(workgroups)), "pwn_tiled_set_reserve")

    # -- sink (screen.h:126-149) ----------------------------------------------
    def screen_upscale(self, sbuf, scale, pitch_bytes=None, pixels=None):
        scale = int(scale)
        if pitch_bytes is None:
            pitch_bytes = self.w * scale * 4
        if pixels is None:
            pixels = np.zeros((self.h * scale, pitch_bytes // 4), np.uint32)
        src = None
        if sbuf is not None:
            sbuf = np.ascontiguousarray(sbuf, np.uint32)
            src = sbuf.ctypes.data
        self._chk(lib.pwn_screen_upscale(self._ctx, src, scale, int(pitch_bytes), pixels.ctypes.data),
                  "pwn_screen_upscale")
        return pixels

    def upscale_device(self, d_src, scale, pitch_bytes, d_dst, stream=0):
        self._chk(lib.pwn_upscale_device(self._ctx, C.c_void_p(d_src), int(scale), int(pitch_bytes),
                                         C.c_void_p(d_dst), C.c_void_p(stream)), "pwn_upscale_device")

    # -- stats / probes --------------------------------------------------------
    def stats(self):
        st = _lib.Stats()
        self._chk(lib.pwn_get_stats(self._ctx, C.byref(st)), "pwn_get_stats")
        out = {n: getattr(st, n) for n, _ in _lib.Stats._fields_ if n != "reserved_"}
        out["wave_paths"] = list(st.wave_paths)
        out["regions"] = list(st.regions)
        return out

    def probe(self, op, words):
        words = np.ascontiguousarray(words).view(np.uint32).ravel()
        per = {_lib.PROBE_DIV: 2, _lib.PROBE_FTOINT: 4}.get(op, 1)
        n = words.size // per
        out = np.zeros(n, np.uint32)
        self._chk(lib.pwn_probe(self._ctx, int(op), words.ctypes.data, out.ctypes.data, n), "pwn_probe")
        return out


# camera helpers the host uses to pose the view (util.h:61-110; outside the
# kernel path, restated for drivers and tests)
def mat4_iden():
    return np.eye(4, dtype=np.float32)


def mat4_roty(m, ang):
    m = np.array(m, np.float32).reshape(4, 4).copy()
    vs, vc = np.float32(np.sin(np.float32(ang))), np.float32(np.cos(np.float32(ang)))
    vxx, vxz, vzx, vzz = m[0, 0], m[0, 2], m[2, 0], m[2, 2]
    m[0, 0] = vc * vxx + vs * vxz
    m[0, 2] = vc * vxz - vs * vxx
    m[2, 0] = vc * vzx + vs * vzz
    m[2, 2] = vc * vzz - vs * vzx
    return m


def mat4_rotx(m, ang):
    m = np.array(m, np.float32).reshape(4, 4).copy()
    vs, vc = np.float32(np.sin(np.float32(ang))), np.float32(np.cos(np.float32(ang)))
    vyy, vyz, vzy, vzz = m[1, 1], m[1, 2], m[2, 1], m[2, 2]
    m[1, 1] = vc * vyy + vs * vyz
    m[1, 2] = vc * vyz - vs * vyy
    m[2, 1] = vc * vzy + vs * vzz
    m[2, 2] = vc * vzz - vs * vzy
    return m


def spawn_camera(spawn, ang_y=0.0, ang_x=0.0):
    """mainloop's camera (main.c:61-64): identity at the spawn cell centre,
    optionally turned like the arrow keys do (main.c:188-193)."""
    cam = mat4_iden()
    if ang_y:
        cam = mat4_roty(cam, ang_y)
    if ang_x:
        cam = mat4_rotx(cam, ang_x)
    cam[3, 0], cam[3, 1], cam[3, 2] = spawn[0] + 0.5, 0.5, spawn[1] + 0.5
    return cam
