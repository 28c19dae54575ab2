"""PWN_OPT_UNIT_ORDER (an option, off by default): the trace kernel hands its units out by what they cost in the last launch of the same rows (every
unit's cost written by the launch, sorted per queue behind the frame's last kernel).  Any order of the units gives the same
frame: every frame of a sequence is the golden frame and has the golden counters, whichever order it was traced in; and the
sorted order really is used (pwn_unit_order_state).  Replaces the static schedule of screen.h:63-64."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, level_path, load_spheres

pytestmark = pytest.mark.gpu


def case(cases, name):
    return [c for c in cases if c["name"] == name][0]


@pytest.mark.parametrize("name", ["level_spawn_1280x720", "synth64_cam1_1920x1080", "synth256_cam0_1920x1080", "level_spawn_3840x2160"])
def test_every_frame_of_a_sequence_is_the_golden_frame_in_any_order(name, cases, oracle_lib):
    import pwnfps_amd
    c = case(cases, name)
    w, h = c["w"], c["h"]
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(level_path(c["level"]))
    r.set_objects(load_spheres(c["spheres"]))
    cam = np.array(c["cam"], np.float32).reshape(4, 4)
    assert r.unit_order_state()["option"] == 0          # off by default: measured a loss on long launches (profiles/r4/unit_order_ab.txt)
    r.set_unit_order(True)
    for k in range(5):
        if k == 3:
            r.set_counters(True)
        sb, z = r.trace_screen_centred(cam, c["sec"])
        assert oracle_lib.fnv64(sb) == c["post"] and oracle_lib.fnv64(z) == c["z"], (name, k)
        if k == 3:
            st = r.stats()
            assert (st["rays"], st["steps"], st["portals"], st["sphere_tests"], st["exhausted"]) == (
                c["rays"], c["steps"], c["portals"], c["sphere_tests"], c["exhausted"])
            r.set_counters(False)
    st = r.unit_order_state()
    assert st["sorts"] == 5 and st["launches_in_sorted_order"] == 4 and st["units_ordered"] == ((w + 15) // 16) * ((h + 3) // 4), st
    # ... and switched off: arithmetic order again, same frame
    r.set_unit_order(False)
    sb, z = r.trace_screen_centred(cam, c["sec"])
    assert oracle_lib.fnv64(sb) == c["post"]
    assert r.unit_order_state()["launches_in_sorted_order"] == 4
    r.close()


def test_order_follows_the_rows_it_was_made_for(cases, oracle_lib):
    """A strip launch (pwn_trace_rows_device) and frames of another camera in between: an order is only used for the rows it was
    sorted from; frames in flight on two streams each keep their stream's own."""
    import pwnfps_amd
    c = case(cases, "level_spawn_1280x720")
    w, h = c["w"], c["h"]
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(level_path(c["level"]))
    r.set_objects(load_spheres(c["spheres"]))
    cam = np.array(c["cam"], np.float32).reshape(4, 4)
    r.set_unit_order(True)
    r.frames_config(3, sbuf=True)
    for k in range(9):
        if k >= 3:
            f = r.wait_frame(k % 3)
            assert oracle_lib.fnv64(f["sbuf"]) == c["post"], k
        r.submit_frame(cam, c["sec"], k % 3)
    for k in range(9, 12):
        f = r.wait_frame(k % 3)
        assert oracle_lib.fnv64(f["sbuf"]) == c["post"], k
    st = r.unit_order_state()
    assert st["sorts"] == 9 and st["launches_in_sorted_order"] == 7, st          # each stream's first frame runs in arithmetic order
    r.frames_config(0)
    sb, _ = r.trace_screen_centred(cam, c["sec"])
    assert oracle_lib.fnv64(sb) == c["post"]
    r.close()


@pytest.mark.parametrize("units", [64, 65, 4 * 64 + 3, 14400, 129600, 518400, 2073600])
def test_the_sort_hands_out_every_unit_once_dearest_first(units):
    """pwn_order_kernel by itself: for random costs (with many ties, zeros and saturated entries) every queue's row is a
    permutation of exactly that queue's units, in order of falling cost -- whatever the costs, every unit is handed out once."""
    import pwnfps_amd
    rng = np.random.default_rng(units)
    cost = rng.integers(0, 40, units).astype(np.uint16) * rng.integers(0, 3, units).astype(np.uint16) * 300
    cost[rng.integers(0, units, max(1, units // 50))] = 65535
    r = pwnfps_amd.Renderer(320, 240)
    perm = r.unit_order_probe(cost)
    r.close()
    cap = (units + 63) // 64
    assert perm.shape == (64, cap)
    seen = np.zeros(units, np.int32)
    for q in range(64):
        n = (units + 63 - q) // 64
        row = perm[q, :n].astype(np.int64)
        assert (perm[q, n:] == 0xffffffff).all()
        assert ((row % 64) == q).all() and (row < units).all()
        np.add.at(seen, row, 1)
        c = np.minimum(cost[row].astype(np.int64) >> 3, 255)          # the sort's cost classes (0.32 us each; order inside one is free)
        assert (np.diff(c) <= 0).all(), q
    assert (seen == 1).all()
