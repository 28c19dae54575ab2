"""game.lua restated for hosts without Lua.

The reference drives its objects from a Lua script through four C callbacks
(script.h:1-64: obj_new, obj_set, obj_free, level_get) and calls the script's
``on_tick(sec_current, sec_delta)`` once per frame (main.c:127-140).  Lua is
not in this image, so the script is restated here in Python over the same four
calls; ``host/game_script.c`` is the same restatement in C over the C ABI.
Lua numbers are doubles and ``math.sin / cos / fmod / floor`` are the C
library's, which is what Python's ``math`` calls too, so the sphere table a
tick produces is the one the reference's VM would hand to obj_set on this libc.

The script's object table (``opos``, game.lua:2-20) is data and lives in
``pwnfps_amd/data/game_objects.txt``: one object per line,
``dx dy dz  r  c1 c2 c3  refl`` (the script's own column order).

``ObjectTable`` is the host-side object list the callbacks act on when no GPU
context is at hand (lv->objs with level_obj_new's slot reuse, level.h:41-62);
``pwnfps_amd.render.Renderer`` offers the same four calls over libpwnhip.so.
"""
import math
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "game_objects.txt")
OBJ_MAX = 10000   # defs.h:4

_SPHERE_DTYPE = np.dtype([("r", "<f4"), ("refl", "<f4"), ("x", "<f4"), ("y", "<f4"),
                          ("z", "<f4"), ("cb", "<f4"), ("cg", "<f4"), ("cr", "<f4")])


def load_object_rows(path=DATA):
    rows = []
    with open(path) as f:
        for line in f:
            line = line.split("#")[0].split()
            if len(line) == 8:
                rows.append([float(v) for v in line])
    return rows


class ObjectTable:
    """lv->objs (defs.h:98-99) with the script callbacks' semantics; no GPU."""

    FREE, INVAL = "free", "inval"

    def __init__(self, cells=None):
        self.slots = []          # per slot: FREE, INVAL or a sphere tuple
        self.cells = cells       # (64,64) uint8, for level_get

    def obj_new(self):
        for i, s in enumerate(self.slots):        # level.h:45-51: first gap
            if s is self.FREE:
                self.slots[i] = self.INVAL
                return i
        if len(self.slots) >= OBJ_MAX:            # level.h:54-55
            raise MemoryError("obj_new: could not allocate object")
        self.slots.append(self.INVAL)
        return len(self.slots) - 1

    def _live(self, obj, who):
        # a handle is a slot level_obj_new once handed out; like the reference's part pointers it
        # stays usable after obj_free (script.h:24 revives a freed part, script.h:48 frees it again)
        if not (0 <= obj < len(self.slots)):
            raise ValueError("%s: %r is not an object" % (who, obj))

    def obj_set(self, obj, typ, r, refl, x, y, z, cb, cg, cr):
        if str(typ).lower() != "sphere":          # script.h:18,34-36 (strcasecmp)
            raise ValueError('obj_set: invalid typ "%s"' % typ)
        self._live(obj, "obj_set")
        self.slots[obj] = tuple(np.float32(v) for v in (r, refl, x, y, z, cb, cg, cr))
        return obj

    def obj_free(self, obj):
        self._live(obj, "obj_free")
        self.slots[obj] = self.FREE

    def level_get(self, cx, cz):                  # script.h:53-64 + util.h:151-158
        cx, cz = int(cx), int(cz)
        if not 0 <= cx < 64:
            cx = 0
        if not 0 <= cz < 64:
            cz = 0
        return chr(int(self.cells[cz][cx]))

    def live(self):
        """What level_prepare_render (level.h:64-81) walks: table order, free slots skipped."""
        rows = []
        for i, s in enumerate(self.slots):
            if s is self.FREE:
                continue
            if s is self.INVAL:
                raise ValueError("object %d was created but never set" % i)   # level.h:34-37 aborts
            rows.append(s)
        return np.array(rows, _SPHERE_DTYPE) if rows else np.zeros(0, _SPHERE_DTYPE)


class GameScript:
    """The shipped game.lua: a cluster of 14 spheres that spins about its
    centre while the centre patrols the level, turning at walls.

    ``host`` provides obj_new / obj_set / level_get (an ``ObjectTable`` or a
    ``Renderer``)."""

    SPD = 2.0                                     # game.lua:63

    def __init__(self, host, rows=None):
        self.host = host
        self.opos = [list(r) for r in (rows if rows is not None else load_object_rows())]
        self.obx, self.oby, self.obz = 9.5, 0.3, 5.5     # game.lua:22
        self.obvx, self.obvz = 1.0, 0.0                  # game.lua:23
        self.oball = []
        for o in self.opos:                              # game.lua:25-30
            h = host.obj_new()
            host.obj_set(h, "sphere", o[3], o[7], self.obx + o[0], self.oby + o[1], self.obz + o[2],
                         o[4], o[5], o[6])
            self.oball.append(h)

    def _blocked(self, c1, c2):                          # game.lua:70,75
        return c2 == "." or ((c1 == "#" or c1 == "&") and c2 == '"')

    def _ahead(self):
        nobx = self.obx + self.obvx * self._dt * self.SPD
        nobz = self.obz + self.obvz * self._dt * self.SPD
        c2 = self.host.level_get(math.floor(nobx + self.obvx * 0.5), math.floor(nobz + self.obvz * 0.5))
        return nobx, nobz, c2

    def on_tick(self, sec_current, sec_delta):
        sec_current, self._dt = float(sec_current), float(sec_delta)
        o2 = self.opos[1]
        if math.fmod(sec_current, 0.5) < 0.15:           # game.lua:36-40: the blinking top sphere
            o2[4], o2[5], o2[6] = 0.3, 0.3, 1.3
        else:
            o2[4], o2[5], o2[6] = 0.3, 0.3, 0.3

        rs = math.sin(sec_current * math.pi * 2 / 2)     # game.lua:48-49
        rc = math.cos(sec_current * math.pi * 2 / 2)
        for h, o in zip(self.oball, self.opos):          # game.lua:42-58
            rx, ry, rz = o[0], o[1], o[2]
            rx, rz = rc * rx + rs * rz, rc * rz - rs * rx
            self.host.obj_set(h, "sphere", o[3], o[7], self.obx + rx, self.oby + ry, self.obz + rz,
                              o[4], o[5], o[6])

        c1 = self.host.level_get(math.floor(self.obx), math.floor(self.obz))   # game.lua:61
        nobx, nobz, c2 = self._ahead()
        if c1 != c2 and self._blocked(c1, c2):           # game.lua:69-82: turn, else turn back
            self.obvx, self.obvz = self.obvz, -self.obvx
            nobx, nobz, c2 = self._ahead()
            if self._blocked(c1, c2):
                self.obvx, self.obvz = -self.obvx, -self.obvz
                nobx, nobz, c2 = self._ahead()
        self.obx, self.obz = nobx, nobz


def frame_times(n, dt):
    """sec_current of frames 0..n-1 for a host that advances a float clock by
    float(dt) per frame (main.c:112-114), and the (sec_current, sec_delta)
    pairs on_tick receives after each frame."""
    t, dt = np.float32(0.0), np.float32(dt)
    secs, ticks = [], []
    for _ in range(n):
        secs.append(float(t))
        t = np.float32(t + dt)
        ticks.append((float(t), float(dt)))
    return secs, ticks
