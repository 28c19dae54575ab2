"""The blocking call in row strips (PWN_OPT_CALL_STRIPS): pwn_trace_screen_centred -- the one call an unchanged
reference loop makes per frame, main.c:107 / screen.h:31-124 -- traces strip k while strip k - 1 is blurred from the
rows traced so far and strip k - 2 travels to the caller's buffers.  Same pixels and depth as one launch per pass,
whatever the strips: goldens of the compiled reference, the oracle, the stale-depth rule, taps that leave the rows
traced so far (repeat over the whole frame), pageable and registered host buffers."""
import os

import numpy as np
import pytest

from conftest import GOLD, level_path, load_spheres

pytestmark = pytest.mark.gpu


def _renderer(w, h):
    import pwnfps_amd
    return pwnfps_amd.Renderer(w, h)


def _case(cases, name):
    return [c for c in cases if c["name"] == name][0]


def _load(r, c):
    r.level_load(level_path(c["level"]))
    r.set_objects(load_spheres(c["spheres"]))
    return np.array(c["cam"], np.float32)


def test_default_runs_large_frames_in_strips_and_small_ones_in_one_piece(oracle_lib, cases):
    c = _case(cases, "level_spawn_3840x2160")
    r = _renderer(c["w"], c["h"])
    cam = _load(r, c)
    post, z = r.trace_screen_centred(cam, c["sec"])
    st = r.call_strips_state()
    assert st["option"] == -1 and st["strips_last"] >= 4 and st["calls_in_strips"] == 1 and st["redone"] == 0, st
    assert oracle_lib.fnv64(post) == c["post"] and oracle_lib.fnv64(z) == c["z"]
    s = r.stats()
    assert s["total_ms"] > 0 and s["trace_ms"] > 0
    # the final frame stays addressable on the device (pwn_screen_upscale(NULL, ...), main.c:108)
    big = r.screen_upscale(None, 1)
    assert (big == post).all()
    # one launch per pass: the same frame
    r.set_call_strips(0)
    post0, z0 = r.trace_screen_centred(cam, c["sec"])
    assert r.call_strips_state()["strips_last"] == 1
    assert (post0 == post).all() and (z0.view(np.uint32) == z.view(np.uint32)).all()
    # without the blur, and without depth wanted
    r.set_call_strips(-1)
    r.set_blur_passes(0)
    pre = r.trace_screen_centred(cam, c["sec"], want_z=False)
    assert r.call_strips_state()["strips_last"] >= 4
    assert oracle_lib.fnv64(pre) == c["pre"]
    assert (r.screen_upscale(None, 1) == pre).all()
    r.close()
    c = _case(cases, "level_spawn_1280x720")
    r = _renderer(c["w"], c["h"])
    cam = _load(r, c)
    post, z = r.trace_screen_centred(cam, c["sec"])
    assert r.call_strips_state()["strips_last"] == 1
    assert oracle_lib.fnv64(post) == c["post"]
    r.close()


@pytest.mark.parametrize("name,strips", [("level_spawn_1280x720", 2), ("level_spawn_1280x720", 7), ("level_pose1_1280x720", 23),
                                         ("synth64_cam2_1920x1080", 5), ("level_spawn_320x240", 3), ("level_spawn_320x240", 32),
                                         ("synth64_cam1_480x272", 4)])
def test_any_number_of_strips_gives_the_golden_frame(oracle_lib, cases, name, strips):
    c = _case(cases, name)
    r = _renderer(c["w"], c["h"])
    cam = _load(r, c)
    r.set_call_strips(strips)
    post, z = r.trace_screen_centred(cam, c["sec"])
    st = r.call_strips_state()
    assert 2 <= st["strips_last"] <= strips, st
    assert oracle_lib.fnv64(post) == c["post"], (name, strips, st)
    assert oracle_lib.fnv64(z) == c["z"], (name, strips)
    r.close()


def test_taps_below_the_rows_traced_so_far_repeat_the_blur(oracle_lib, cases):
    """synth256: halls of mirrors, depths in the hundreds -- the blur's taps reach far below a chunk; the call notices, repeats the
    pass over the whole frame and hands over the exact frame"""
    c = _case(cases, "synth256_cam0_1920x1080")
    r = _renderer(c["w"], c["h"])
    cam = _load(r, c)
    r.set_call_strips(8)
    post, z = r.trace_screen_centred(cam, c["sec"])
    st = r.call_strips_state()
    assert st["strips_last"] >= 2 and st["redone"] == 1, st
    assert oracle_lib.fnv64(post) == c["post"] and oracle_lib.fnv64(z) == c["z"]
    r.close()


def test_after_a_repeat_the_next_calls_run_in_one_piece():
    """by frame size (the default): a frame whose blur had to be repeated costs more than one launch per pass, so the calls behind it
    cover deeper taps and then run in one piece for a while.  Two contexts, the same six frames (this scene's rays run out of steps: depth carries over
    from frame to frame), one by size into a registered buffer, one with one launch per pass: the same frames."""
    w, h = 2560, 1440
    cams = np.load(os.path.join(GOLD, "levels", "synth256_cams.npy"))
    sph = load_spheres("synth256")
    out = []
    for strips in (-1, 0):
        r = _renderer(w, h)
        r.level_load(level_path("synth256"))
        r.set_objects(sph)
        r.set_call_strips(strips)
        sb = np.zeros((h, w), np.uint32)
        zb = np.zeros((h, w), np.float32)
        r.host_register(sb)
        r.host_register(zb)
        frames = []
        seen = []
        for i, sec in enumerate((0.0, 0.5, 1.0, 1.5, 2.0, 2.5)):
            r.trace_screen_centred(cams[i % 2], sec, sbuf=sb, zbuf=zb)
            frames.append((sb.copy(), zb.copy()))
            st = r.call_strips_state()
            seen.append((st["strips_last"], st["redone"]))
        if strips < 0:
            # strips blurred where taps of depth 8 are covered until a frame's taps go further (repeat), then depth 24 until that is not
            # enough either (repeat), then in one piece: two repeats in all, and the calls behind the second run in one piece
            assert seen[0][0] >= 4 and st["redone"] == 2 and seen[-1][0] == 1, seen
            second = [i for i, (_, red) in enumerate(seen) if red == 2][0]
            assert all(k == 1 for k, _ in seen[second + 1:]) and all(k >= 4 for k, _ in seen[:second + 1]), seen
        r.host_unregister(sb)
        r.host_unregister(zb)
        r.close()
        out.append(frames)
    for (a, za), (b, zb_) in zip(out[0], out[1]):
        assert (a == b).all() and (za.view(np.uint32) == zb_.view(np.uint32)).all()


def test_stale_depth_semantics_through_strips(oracle_lib):
    """zbuf keeps its previous value where the primary ray exhausts maxsteps (trace.h:677), strip by strip as in one launch"""
    cams = np.load(os.path.join(GOLD, "levels", "synth256_cams.npy"))
    sph = load_spheres("synth256")
    O = oracle_lib.Oracle()
    O.load_level(level_path("synth256"))
    O.set_spheres(sph)
    w, h = 480, 272
    for blur in (0, 1):
        r = _renderer(w, h)
        r.level_load(level_path("synth256"))
        r.set_objects(sph)
        r.set_blur_passes(blur)
        r.set_call_strips(5)
        _, z1 = r.trace_screen_centred(cams[1], 0.0)
        a, z2 = r.trace_screen_centred(cams[0], 0.0)
        assert r.call_strips_state()["calls_in_strips"] == 2
        sb, zb, st = O.trace_rows(w, h, 0, h, cams[1])
        sb, zb, st = O.trace_rows(w, h, 0, h, cams[0], sb=sb, zb=zb)
        assert st.exhausted > 0
        assert (z2.view(np.uint32) == zb.view(np.uint32)).all()
        assert (a == (O.blur_rows(0, h, sb, zb) if blur else sb)).all()
        r.close()


def test_registered_and_pageable_host_buffers(oracle_lib, cases):
    c = _case(cases, "level_spawn_1920x1080")
    r = _renderer(c["w"], c["h"])
    cam = _load(r, c)
    r.set_call_strips(6)
    sb = np.zeros((c["h"], c["w"]), np.uint32)
    zb = np.zeros((c["h"], c["w"]), np.float32)
    r.host_register(sb)
    r.host_register(zb)
    r.host_register(sb)                                # (again: the same range, fine)
    r.trace_screen_centred(cam, c["sec"], sbuf=sb, zbuf=zb)
    assert oracle_lib.fnv64(sb) == c["post"] and oracle_lib.fnv64(zb) == c["z"]
    sb[:] = 0
    r.trace_screen_centred(cam, c["sec"], sbuf=sb, want_z=False)
    assert oracle_lib.fnv64(sb) == c["post"]
    r.host_unregister(zb)
    r.host_unregister(sb)
    with pytest.raises(Exception):
        r.host_unregister(sb)
    sb[:] = 0
    zb[:] = 0
    r.trace_screen_centred(cam, c["sec"], sbuf=sb, zbuf=zb)          # pageable again
    assert oracle_lib.fnv64(sb) == c["post"] and oracle_lib.fnv64(zb) == c["z"]
    r.close()


def test_strips_mixed_with_frames_in_flight(oracle_lib, cases):
    """the blocking call is ordered behind the frames in flight, in strips too"""
    c = _case(cases, "level_spawn_1920x1080")
    r = _renderer(c["w"], c["h"])
    cam = _load(r, c)
    r.set_call_strips(4)
    r.frames_config(3, sbuf=True)
    for i in range(3):
        r.submit_frame(cam, c["sec"], i)
    post, z = r.trace_screen_centred(cam, c["sec"])
    assert oracle_lib.fnv64(post) == c["post"] and oracle_lib.fnv64(z) == c["z"]
    for i in range(3):
        f = r.wait_frame(i)
        assert oracle_lib.fnv64(f["sbuf"]) == c["post"]
    r.submit_frame(cam, c["sec"], 0)
    assert oracle_lib.fnv64(r.wait_frame(0)["sbuf"]) == c["post"]
    r.frames_config(0)
    r.close()


@pytest.mark.parametrize("lists", ["indexed", "inline"])
def test_both_forms_of_the_sphere_lists_give_the_goldens(oracle_lib, cases, monkeypatch, lists):
    """the per-cell sphere lists (trace.h:252-296) as u16 indices into the sphere array or with the sphere records inline (tables.h, round 5):
    chosen by the size of the tables, here forced each way -- every level, both lane forms, counters included"""
    monkeypatch.setenv("PWN_SPHERE_LISTS", lists)
    names = ("level_spawn_1280x720", "level_pose1_1280x720", "synth64_cam1_1920x1080", "synth256_cam1_480x272", "level_spawn_320x200", "level_spawn_nosph_320x240")
    for hasw in (False, True):
        if hasw:
            monkeypatch.setenv("PWN_DBG_FORCE_HASW", "1")
        for name in names:
            c = _case(cases, name)
            r = _renderer(c["w"], c["h"])
            cam = _load(r, c)
            r.set_counters(True)
            post, z = r.trace_screen_centred(cam, c["sec"])
            st = r.stats()
            assert oracle_lib.fnv64(post) == c["post"] and oracle_lib.fnv64(z) == c["z"], (lists, hasw, name)
            if "steps" in c:
                assert (st["rays"], st["steps"], st["portals"], st["sphere_tests"], st["exhausted"]) == (c["rays"], c["steps"], c["portals"], c["sphere_tests"], c["exhausted"]), (lists, hasw, name)
            r.set_counters(False)
            post2, _ = r.trace_screen_centred(cam, c["sec"])
            assert (post2 == post).all()
            # the other scheduler reads indexed lists: the tables are packed again behind the option
            r.set_scheduler("refill")
            post3, _ = r.trace_screen_centred(cam, c["sec"])
            assert (post3 == post).all(), (lists, hasw, name)
            r.close()


def test_a_context_finds_out_about_its_copy_streams_once(oracle_lib, cases, monkeypatch):
    """chunks go to the host on one copy stream or on two in turn -- which is faster depends on the queues the runtime handed the context --
    so the first 8 calls in strips use one, the next 8 two, and the faster stays; every call delivers the golden frame meanwhile"""
    c = _case(cases, "level_spawn_1920x1080")
    r = _renderer(c["w"], c["h"])
    cam = _load(r, c)
    r.set_call_strips(6)
    sb = np.zeros((c["h"], c["w"]), np.uint32)
    r.host_register(sb)
    for i in range(18):
        r.trace_screen_centred(cam, c["sec"], want_z=False, sbuf=sb)
        assert oracle_lib.fnv64(sb) == c["post"], i
        st = r.call_strips_state()
        assert (st["copy_streams"] == 0) == (i < 15) and st["copy_streams"] in (0, 1, 2), (i, st)
    assert st["reach_depth"] == 8 and st["redone"] == 0
    r.host_unregister(sb)
    r.close()
    monkeypatch.setenv("PWN_CALL_COPY_STREAMS", "2")
    r = _renderer(c["w"], c["h"])
    cam = _load(r, c)
    r.set_call_strips(6)
    post, _ = r.trace_screen_centred(cam, c["sec"])
    assert r.call_strips_state()["copy_streams"] == 2 and oracle_lib.fnv64(post) == c["post"]
    r.close()
