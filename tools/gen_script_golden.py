#!/usr/bin/env python3
"""Generate tests/golden/script_ticks.npz: the reference's OWN game script, executed.

The image has no Lua, so game.lua's text (given by path; /root/reference/game.lua in the build
container) is run by tools/minilua.py, a generic interpreter for the subset of Lua 5.1 the
script uses, with the four host callbacks of script.h:1-64 bound to a plain object table:

    obj_new()                           level_obj_new's slot rule (level.h:41-62)
    obj_set(o, "sphere", r, refl, x, y, z, b, g, r)   doubles narrowed to float on store (script.h:22-32)
    obj_free(o)
    level_get(cx, cz)                   lua_tointeger of both, get_cell's clamp (util.h:151-158)

and ticked the way mainloop does (main.c:112-140: a float clock, on_tick(sec_current, tdiff)).
What is written is data: per tick the clock, the sphere table the script left in the object
table, and its globals obx / obz / obvx / obvz.  The restatements of the script's logic
(pwnfps_amd/script.py, host/game_script.c) are held against it by tests/test_script.py.

    python3 tools/gen_script_golden.py [path/to/game.lua]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import minilua  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
SPH = np.dtype([("r", "<f4"), ("refl", "<f4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("cb", "<f4"), ("cg", "<f4"), ("cr", "<f4")])


class Host:
    """lv->objs and lv->data behind the script's callbacks"""
    FREE, INVAL = "free", "inval"

    def __init__(self, cells):
        self.slots, self.cells = [], cells

    def obj_new(self):
        for i, s in enumerate(self.slots):
            if s is self.FREE:
                self.slots[i] = self.INVAL
                return float(i)
        self.slots.append(self.INVAL)
        return float(len(self.slots) - 1)

    def obj_set(self, o, typ, r, refl, x, y, z, cb, cg, cr):
        if o is None:
            raise minilua.LuaError("obj_set: pt cannot be nil")
        if typ is None or str(typ).lower() != "sphere":
            raise minilua.LuaError('obj_set: invalid typ "%s"' % typ)
        self.slots[int(o)] = tuple(np.float32(v) for v in (r, refl, x, y, z, cb, cg, cr))
        return o

    def obj_free(self, o):
        self.slots[int(o)] = self.FREE

    def level_get(self, cx, cz):
        cx, cz = int(cx), int(cz)          # lua_tointeger
        if not 0 <= cx < 64:
            cx = 0
        if not 0 <= cz < 64:
            cz = 0
        return chr(int(self.cells[cz][cx]))

    def table(self):
        return np.array([s for s in self.slots if s is not self.FREE], SPH)


def run(src, cells, n, dt):
    h = Host(cells)
    L = minilua.Interp()
    for name in ("obj_new", "obj_set", "obj_free", "level_get"):
        L.register(name, getattr(h, name))
    L.register("level_set", lambda *a: None)            # script.h:65-69 is a stub
    L.run(src)                                           # script_newvm: the chunk creates the objects
    t, dtf = np.float32(0.0), np.float32(dt)
    secs, tabs, centre = [], [], []
    g = L.globals
    for _ in range(n):
        secs.append(float(t))
        tabs.append(h.table())
        centre.append((g.get("obx"), g.get("obz"), g.get("obvx"), g.get("obvz")))
        t = np.float32(t + dtf)                          # main.c:112-114
        L.call("on_tick", float(t), float(dtf))          # main.c:127-140
    return np.array(secs, np.float32), np.stack(tabs), np.array(centre, np.float64)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/game.lua"
    src = open(path).read()
    cells = np.load(os.path.join(G, "levels", "pwnfps_level_tables.npz"))["data"]
    save = {}
    # the two runs of tests/golden/anim.npz, and a long patrol at 60 Hz that turns at walls in every direction
    for name, n, dt in (("static", 16, 0.05), ("chase", 24, 0.25), ("patrol", 1500, 1.0 / 60.0)):
        secs, tabs, centre = run(src, cells, n, dt)
        save[name + "_dt"] = np.float32(dt)
        save[name + "_sec"] = secs
        save[name + "_spheres"] = tabs
        save[name + "_centre"] = centre
        print(name, n, "ticks;", len({tuple(c[2:]) for c in centre}), "headings; sphere table", tabs.shape)
    np.savez_compressed(os.path.join(G, "script_ticks.npz"), **save)


if __name__ == "__main__":
    main()
