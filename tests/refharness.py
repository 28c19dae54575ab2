"""ctypes wrapper over oracle/_ref/libpwnref_*.so (the reference's own headers
compiled by oracle/Makefile).  TEST INFRASTRUCTURE: imported only by tests/,
tools/gen_goldens.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

SPHERE_DTYPE = np.dtype([("r", "<f4"), ("refl", "<f4"), ("x", "<f4"), ("y", "<f4"),
                         ("z", "<f4"), ("cb", "<f4"), ("cg", "<f4"), ("cr", "<f4")])


def available(variant="tab"):
    return os.path.exists(os.path.join(REF_DIR, "libpwnref_%s.so" % variant))


def fnv64(a):
    """SURVEY.md App. B6 frame hash over uint32 words, row-major."""
    a = np.ascontiguousarray(a).view(np.uint32).ravel()
    h = 1469598103934665603
    # vectorising FNV is not possible (sequential); use a chunked python loop
    # via int arithmetic on a bytes view -- fine for <= a few Mpixel in tests.
    m = (1 << 64) - 1
    for p in a.tolist():
        h = ((h ^ p) * 1099511628211) & m
    return "%016x" % h


class RefHarness:
    def __init__(self, variant="tab"):
        path = os.path.join(REF_DIR, "libpwnref_%s.so" % variant)
        self.lib = C.CDLL(path)
        L = self.lib
        L.pwnref_load_level.argtypes = [C.c_char_p]
        L.pwnref_render.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int,
                                    C.c_int, C.c_void_p, C.c_void_p]
        L.pwnref_set_spheres.argtypes = [C.c_void_p, C.c_int]
        L.pwnref_get_level.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.pwnref_set_level.argtypes = [C.c_void_p, C.c_void_p]
        L.pwnref_get_bins.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.pwnref_upscale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.pwnref_col_ftoint.argtypes = [C.c_void_p]
        L.pwnref_col_ftoint.restype = C.c_uint32
        L.pwnref_normalise.argtypes = [C.c_void_p, C.c_void_p]
        L.pwnref_dot.argtypes = [C.c_void_p, C.c_void_p]
        L.pwnref_dot.restype = C.c_float
        for n in ("rcp", "rsqrt", "sinf", "cosf", "expf"):
            f = getattr(L, "pwnref_" + n)
            f.argtypes = [C.c_float]
            f.restype = C.c_float
        for n in ("randfs", "randfu"):
            f = getattr(L, "pwnref_" + n)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_float
        L.pwnref_randi.argtypes = [C.c_void_p]
        L.pwnref_randi.restype = C.c_uint32
        L.pwnref_get_cell.argtypes = [C.c_int, C.c_int]
        L.pwnref_mat4_roty.argtypes = [C.c_void_p, C.c_float]
        L.pwnref_mat4_rotx.argtypes = [C.c_void_p, C.c_float]
        self.variant = L.pwnref_variant()
        if self.variant & 4:
            L.pwnref_get_counters.argtypes = [C.c_void_p]

    def load_level(self, path):
        r = self.lib.pwnref_load_level(path.encode())
        if r != 0:
            raise RuntimeError("pwnref_load_level(%s) -> %d" % (path, r))

    def get_level(self):
        data = np.zeros((64, 64), np.uint8)
        pmap = np.zeros((26, 7), np.int32)
        spawn = np.zeros(2, np.int32)
        r = self.lib.pwnref_get_level(data.ctypes.data, pmap.ctypes.data, spawn.ctypes.data)
        assert r == 0
        return data, pmap, spawn

    def set_level(self, data, pmap):
        data = np.ascontiguousarray(data, np.uint8)
        pmap = np.ascontiguousarray(pmap, np.int32)
        assert data.shape == (64, 64) and pmap.shape == (26, 7)
        self.lib.pwnref_set_level(data.ctypes.data, pmap.ctypes.data)

    def set_spheres(self, sph):
        sph = np.ascontiguousarray(sph, SPHERE_DTYPE)
        r = self.lib.pwnref_set_spheres(sph.ctypes.data, len(sph))
        assert r == 0, r

    def get_bins(self, cap=1 << 20):
        counts = np.zeros(4096, np.uint16)
        idx = np.zeros(cap, np.int32)
        k = self.lib.pwnref_get_bins(counts.ctypes.data, idx.ctypes.data, cap)
        assert k >= 0, k
        return counts, idx[:k].copy()

    def render(self, w, h, cam, sec=0.0, blur=1, threads=0, want_z=True):
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        sb = np.zeros((h, w), np.uint32)
        zb = np.zeros((h, w), np.float32) if want_z else None
        r = self.lib.pwnref_render(w, h, cam.ctypes.data, sec, threads, blur,
                                   sb.ctypes.data, zb.ctypes.data if want_z else None)
        if r != 0:
            raise RuntimeError("pwnref_render -> %d" % r)
        return sb, zb

    def counters(self):
        out = np.zeros(8, np.int64)
        self.lib.pwnref_get_counters(out.ctypes.data)
        return out

    def reset_counters(self):
        self.lib.pwnref_reset_counters()

    def upscale(self, src, scale, pitch_bytes=None):
        h, w = src.shape
        if pitch_bytes is None:
            pitch_bytes = w * scale * 4
        dst = np.zeros((h * scale, pitch_bytes // 4), np.uint32)
        src = np.ascontiguousarray(src, np.uint32)
        self.lib.pwnref_upscale(src.ctypes.data, w, h, scale, pitch_bytes, dst.ctypes.data)
        return dst


def identity_cam(x, y, z):
    """mainloop's initial camera (main.c:61-64): identity at (x,y,z)."""
    cam = np.eye(4, dtype=np.float32)
    cam[3, 0], cam[3, 1], cam[3, 2] = x, y, z
    return cam


def game_lua_spheres():
    """The 14 spheres game.lua:2-30 creates at load (t=0): obj_set(r, refl,
    obx+dx, oby+dy, obz+dz, c1, c2, c3) with (obx,oby,obz)=(9.5,0.3,5.5);
    sums are Lua doubles, narrowed to float by script.h:22-32.  The script's
    object table is data: pwnfps_amd/data/game_objects.txt."""
    sys.path.insert(0, ROOT)
    from pwnfps_amd.script import load_object_rows
    opos = load_object_rows()
    s = np.zeros(len(opos), SPHERE_DTYPE)
    for i, (dx, dy, dz, r, c1, c2, c3, refl) in enumerate(opos):
        s[i] = (r, refl, 9.5 + dx, 0.3 + dy, 5.5 + dz, c1, c2, c3)
    return s
