"""ctypes wrapper over oracle/libpwnoracle.so (the CPU restatement of the
reference's hot path).  TEST INFRASTRUCTURE: imported only by tests/, tools/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by pwnfps_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libpwnoracle.so")

SPHERE_DTYPE = np.dtype([("r", "<f4"), ("refl", "<f4"), ("x", "<f4"), ("y", "<f4"),
                         ("z", "<f4"), ("cb", "<f4"), ("cg", "<f4"), ("cr", "<f4")])


class Stats(C.Structure):
    _fields_ = [("rays", C.c_int64), ("steps", C.c_int64), ("portals", C.c_int64),
                ("sphere_tests", C.c_int64), ("exhausted", C.c_int64)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.pwno_level_new.restype = C.c_void_p
        L.pwno_level_free.argtypes = [C.c_void_p]
        L.pwno_level_load_file.argtypes = [C.c_void_p, C.c_char_p]
        L.pwno_level_load_mem.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.pwno_level_set_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.pwno_level_set_spheres.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.pwno_trace_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pwno_blur_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.pwno_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int,
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pwno_upscale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pwno_col_ftoint.argtypes = [C.c_void_p]
        L.pwno_col_ftoint.restype = C.c_uint32
        L.pwno_normalise.argtypes = [C.c_void_p, C.c_void_p]
        L.pwno_dot.argtypes = [C.c_void_p, C.c_void_p]
        L.pwno_dot.restype = C.c_float
        for n in ("rcp", "rsqrt", "sinf", "cosf", "expf"):
            f = getattr(L, "pwno_" + n)
            f.argtypes = [C.c_float]
            f.restype = C.c_float
        for n in ("randfs", "randfu"):
            f = getattr(L, "pwno_" + n)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_float
        L.pwno_randi.argtypes = [C.c_void_p]
        L.pwno_randi.restype = C.c_uint32
        L.pwno_pixel_seed.argtypes = [C.c_int, C.c_int, C.c_int]
        L.pwno_pixel_seed.restype = C.c_uint32
        L.pwno_blur_seed_at.argtypes = [C.c_int, C.c_int]
        L.pwno_blur_seed_at.restype = C.c_uint32
        L.pwno_frame_setup.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pwno_fnv64.argtypes = [C.c_void_p, C.c_int64]
        L.pwno_fnv64.restype = C.c_uint64
        L.pwno_mat4_roty.argtypes = [C.c_void_p, C.c_float]
        L.pwno_mat4_rotx.argtypes = [C.c_void_p, C.c_float]
        _lib = L
    return _lib


def fnv64(a):
    a = np.ascontiguousarray(a).view(np.uint32).ravel()
    return "%016x" % lib().pwno_fnv64(a.ctypes.data, a.size)


class LevelStruct(C.Structure):
    _fields_ = [("data", C.c_uint8 * 4096), ("pmap", C.c_int32 * (26 * 7)),
                ("sx", C.c_int32), ("sz", C.c_int32), ("nspheres", C.c_int32),
                ("spheres", C.c_void_p), ("bin_off", C.c_int32 * 4097),
                ("bin_idx", C.POINTER(C.c_int32)), ("bin_cap", C.c_int32)]


class Oracle:
    """One level + sphere set; mirrors tests/refharness.RefHarness."""

    def __init__(self):
        self.L = lib()
        self.lv = self.L.pwno_level_new()

    def __del__(self):
        try:
            self.L.pwno_level_free(self.lv)
        except Exception:
            pass

    def _s(self):
        return LevelStruct.from_address(self.lv)

    def load_level(self, path):
        r = self.L.pwno_level_load_file(self.lv, path.encode())
        if r != 0:
            raise RuntimeError("pwno_level_load_file(%s) -> %d" % (path, r))

    def load_level_text(self, text):
        if isinstance(text, str):
            text = text.encode("latin-1")
        assert self.L.pwno_level_load_mem(self.lv, text, len(text)) == 0

    def get_level(self):
        s = self._s()
        data = np.frombuffer(s.data, np.uint8).reshape(64, 64).copy()
        pmap = np.frombuffer(s.pmap, np.int32).reshape(26, 7).copy()
        return data, pmap, np.array([s.sx, s.sz], np.int32)

    def set_level(self, data, pmap):
        data = np.ascontiguousarray(data, np.uint8)
        pmap = np.ascontiguousarray(pmap, np.int32)
        assert data.shape == (64, 64) and pmap.shape == (26, 7)
        self.L.pwno_level_set_tables(self.lv, data.ctypes.data, pmap.ctypes.data)

    def set_spheres(self, sph):
        sph = np.ascontiguousarray(sph, SPHERE_DTYPE)
        assert self.L.pwno_level_set_spheres(self.lv, sph.ctypes.data, len(sph)) == 0

    def get_bins(self):
        s = self._s()
        off = np.frombuffer(s.bin_off, np.int32).copy()
        n = int(off[4096])
        idx = np.ctypeslib.as_array(s.bin_idx, shape=(max(n, 1),))[:n].copy()
        return np.diff(off).astype(np.uint16), idx

    def render(self, w, h, cam, sec=0.0, blur=1, threads=0, want_z=True, stats=False):
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        sb = np.zeros((h, w), np.uint32)
        zb = np.zeros((h, w), np.float32)
        st = Stats()
        r = self.L.pwno_render(self.lv, w, h, cam.ctypes.data, sec, blur, threads,
                               sb.ctypes.data, zb.ctypes.data, C.byref(st))
        if r != 0:
            raise RuntimeError("pwno_render -> %d" % r)
        if stats:
            return sb, zb, st
        return sb, zb

    def trace_rows(self, w, h, y0, y1, cam, sec=0.0, threads=0, sb=None, zb=None):
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        if sb is None:
            sb = np.zeros((h, w), np.uint32)
        if zb is None:
            zb = np.zeros((h, w), np.float32)
        st = Stats()
        r = self.L.pwno_trace_rows(self.lv, w, h, y0, y1, cam.ctypes.data, sec, threads,
                                   sb.ctypes.data, zb.ctypes.data, C.byref(st))
        assert r == 0, r
        return sb, zb, st

    def blur_rows(self, y0, y1, pre, zb, out=None, threads=0):
        h, w = pre.shape
        pre = np.ascontiguousarray(pre, np.uint32)
        zb = np.ascontiguousarray(zb, np.float32)
        if out is None:
            out = pre.copy()
        r = self.L.pwno_blur_rows(w, h, y0, y1, threads, pre.ctypes.data, zb.ctypes.data, out.ctypes.data)
        assert r == 0, r
        return out

    def upscale(self, src, scale, pitch_bytes=None):
        h, w = src.shape
        if pitch_bytes is None:
            pitch_bytes = w * scale * 4
        dst = np.zeros((h * scale, pitch_bytes // 4), np.uint32)
        src = np.ascontiguousarray(src, np.uint32)
        assert self.L.pwno_upscale(src.ctypes.data, w, h, scale, pitch_bytes, dst.ctypes.data) == 0
        return dst
