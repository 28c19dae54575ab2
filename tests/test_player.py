"""host/player.c: the interactive half of the reference's loop (main.c:142-379: key state, turning,
walking with push-back, gravity, the step between the levels of a two-high room, walking through
a portal) restated for hosts over the C ABI.  PARITY UNPINNED by the reference (main.c needs SDL
and Lua, and ships no fixtures).  What can be checked without it:
  * its own invariants on the shipped level (the camera never enters a solid cell, rests 0.3 from walls,
    settles at y = 0.4),
  * consistency with the PINNED ray path: walking through a portal must shorten the view ray --
    traced by the oracle through the same portal -- by exactly the distance walked, and leave the
    picture's centre unchanged in kind,
  * the key-script format."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, ROOT, level_path


class Keys(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("turnleft", "turnright", "turnup", "turndown", "moveforward", "moveback", "moveleft", "moveright")]


class Player(C.Structure):
    _fields_ = [("cam", C.c_float * 16), ("gravity", C.c_float * 4), ("traversals", C.c_int)]


class KeyEvent(C.Structure):
    _fields_ = [("frame", C.c_int), ("sym", C.c_int), ("down", C.c_int)]


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), os.path.join(ROOT, "host", "libpwnplayer.so")])
    L = C.CDLL(os.path.join(ROOT, "host", "libpwnplayer.so"))
    L.pwn_player_init.argtypes = [C.POINTER(Player), C.c_void_p]
    L.pwn_player_step.argtypes = [C.POINTER(Player), C.POINTER(Keys), C.c_float, C.c_void_p, C.c_void_p]
    L.pwn_keys_event.argtypes = [C.POINTER(Keys), C.c_int, C.c_int]
    L.pwn_key_from_name.argtypes = [C.c_char_p]
    L.pwn_keys_load.argtypes = [C.c_char_p, C.POINTER(KeyEvent), C.c_int]
    return L


@pytest.fixture(scope="module")
def tables():
    t = np.load(os.path.join(GOLD, "levels", "pwnfps_level_tables.npz"))
    return np.ascontiguousarray(t["data"], np.uint8), np.ascontiguousarray(t["pmap"], np.int32), np.ascontiguousarray(t["spawn"], np.int32)


def fresh(lib, tables, pos=None):
    data, pmap, spawn = tables
    p = Player()
    lib.pwn_player_init(C.byref(p), spawn.ctypes.data)
    if pos is not None:
        p.cam[12], p.cam[13], p.cam[14] = pos
    return p


def step(lib, tables, p, k, dt=1.0 / 60.0, n=1):
    data, pmap, _ = tables
    for _ in range(n):
        lib.pwn_player_step(C.byref(p), C.byref(k), dt, data.ctypes.data, pmap.ctypes.data)


SOLID_FOR_FEET = set('.')           # at y in [0,1): everything that is not a room, ramp or paired portal


def test_start_pose_gravity_and_rest(lib, tables):
    p = fresh(lib, tables)
    assert list(p.cam) == [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 9.5, 0.5, 4.5, 1]          # main.c:59-64
    k = Keys()
    ys = []
    for _ in range(40):
        step(lib, tables, p, k)
        ys.append(p.cam[13])
    assert ys[0] == 0.5 and all(a >= b for a, b in zip(ys, ys[1:]))          # main.c:268-270: gravity starts at 0
    # main.c:272-276: on the floor; the pull builds up for one tick and is cancelled on the next
    assert ys[-5:] == [np.float32(0.4)] * 5 and p.gravity[1] in (0.0, -np.float32(3.0) * np.float32(1 / 60.0) * np.float32(1 / 60.0))
    assert (p.cam[12], p.cam[14]) == (9.5, 4.5)


def test_walls_push_back(lib, tables):
    data = tables[0]
    p = fresh(lib, tables)
    k = Keys(moveforward=1)
    # forward is +z: (9,4) ';' -> (9,5) '#' -> (9,6) '#' -> (9,7) '.' is solid
    assert [chr(data[z, 9]) for z in (4, 5, 6, 7)] == [';', '#', '#', '.']
    step(lib, tables, p, k, n=200)
    assert p.cam[14] == np.float32(6 + 0.5 + np.float32(0.5 - np.float32(0.2))) and p.cam[12] == 9.5      # main.c:258-266
    # sliding along the wall: forward + left (a = +x of the right-hand row) moves in x only
    k = Keys(moveforward=1, moveleft=1)
    x0 = p.cam[12]
    step(lib, tables, p, k, n=3)
    assert p.cam[12] > x0 and p.cam[14] == np.float32(6.8)
    # a long random walk never ends up in a solid cell
    rng = np.random.default_rng(7)
    p = fresh(lib, tables)
    k = Keys()
    for i in range(6000):
        if i % 20 == 0:
            k = Keys(*[int(v) for v in rng.integers(0, 2, 8)])
        step(lib, tables, p, k)
        c = chr(data[int(p.cam[14]), int(p.cam[12])])
        assert c not in SOLID_FOR_FEET, (i, c, list(p.cam[12:15]))
    assert p.traversals > 0                                                   # it found portals on the way


def _depth_ahead(oracle_lib, cam, w=16):
    """the oracle's depth of the ray along the camera's forward row (pixel (w/2-1, w/2) of a w x w frame)"""
    from oracle import Oracle
    O = Oracle()
    O.load_level(level_path("pwnfps_level"))
    O.set_spheres(np.zeros(0, oracle_lib.SPHERE_DTYPE))
    img, z = O.render(w, w, np.array(cam, np.float32).reshape(4, 4), sec=0.0, blur=0)
    return float(z[w // 2, w // 2 - 1]), int(img[w // 2, w // 2 - 1])


# F (14,5)<->(9,12) keeps the heading (walked both ways), C (22,7)->(10,8) too (entered going -z),
# K (6,3)->(2,5) turns the traveller by a quarter (rot12 = 3): in going +z, out going +x
@pytest.mark.parametrize("start,turn,letter", [((8.5, 0.4, 12.5), np.pi / 2, "F"), ((15.5, 0.4, 5.5), -np.pi / 2, "F"),
                                                ((22.5, 0.4, 8.5), np.pi, "C"), ((6.5, 0.4, 2.5), 0.0, "K")])
def test_walking_through_a_portal_agrees_with_the_ray_path(lib, tables, oracle_lib, start, turn, letter):
    """The ray's portal crossing is pinned (trace.h:508-650 against the compiled reference).  The player's
    (main.c:284-378) must be the same map: a walk of s units straight ahead shortens the oracle's view
    ray by s, whether or not a portal lies in between -- also for the portals that turn the traveller."""
    data, pmap, _ = tables
    p = fresh(lib, tables, start)
    k = Keys(turnleft=1)
    lib.pwn_player_step(C.byref(p), C.byref(k), np.float32(turn / 3.0), data.ctypes.data, pmap.ctypes.data)   # one tick that only turns
    p.cam[12], p.cam[13], p.cam[14] = start
    p.gravity[1] = 0.0
    d0, c0 = _depth_ahead(oracle_lib, p.cam)
    k = Keys(moveforward=1)
    walked, seen = 0.0, set()
    dt = np.float32(1.0 / 120.0)
    for i in range(400):
        before = (p.cam[12], p.cam[14], p.traversals)
        step(lib, tables, p, k, dt)
        walked += float(dt) * 5.0
        if p.traversals != before[2]:
            seen.add(letter)
        d, c = _depth_ahead(oracle_lib, p.cam)
        if d0 - walked < 0.45:              # close to the far wall: the bounding box stops the player
            break
        assert abs(d - (d0 - walked)) < 2e-3, (i, d, d0 - walked, list(p.cam))
    assert seen == {letter}, "the walk was meant to cross portal " + letter


def test_key_script(lib, tmp_path):
    path = tmp_path / "keys.txt"
    path.write_text("# frame key state\n0 w down\n3 left down   # turn while walking\n5 left up\n9 w up\n12 quit down\n")
    ev = (KeyEvent * 16)()
    n = lib.pwn_keys_load(str(path).encode(), ev, 16)
    assert n == 5
    assert [(e.frame, e.sym, e.down) for e in ev[:n]] == [(0, 4, 1), (3, 0, 1), (5, 0, 0), (9, 4, 0), (12, 8, 1)]
    k = Keys()
    lib.pwn_keys_event(C.byref(k), 4, 1)
    lib.pwn_keys_event(C.byref(k), 0, 1)
    assert (k.moveforward, k.turnleft, k.moveback) == (1, 1, 0)
    lib.pwn_keys_event(C.byref(k), 4, 0)
    assert k.moveforward == 0
    for bad in ("1 jump down\n", "x w down\n", "2 w sideways\n", "3 w\n"):
        path.write_text(bad)
        assert lib.pwn_keys_load(str(path).encode(), ev, 16) == -1
    assert lib.pwn_keys_load(b"/nonexistent", ev, 16) == -1
    assert lib.pwn_key_from_name(b"d") == 7 and lib.pwn_key_from_name(b"space") == 9
