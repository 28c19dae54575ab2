"""Frames in flight (pwn_frames_config / pwn_submit_frame / pwn_wait_frame): the loop of
main.c:93-109 with the hand-over to the host overlapped.  Every delivered frame -- colour,
depth and upscaled surface -- is the oracle's frame for the inputs of ITS submit, although
spheres, camera and sec_current change between submits without any wait."""
import os

import numpy as np
import pytest

from conftest import GOLD, level_path, load_spheres

pytestmark = pytest.mark.gpu

W, H = 640, 352


def _scene(f, base):
    """frame f: a turned camera, moved and recoloured spheres, one sphere fewer every third frame"""
    import pwnfps_amd
    sph = base.copy()
    sph["x"] += np.float32(0.11 * f)
    sph["z"] -= np.float32(0.07 * f)
    sph["cb"] = np.float32(0.2 + 0.1 * (f % 5))
    if f % 3 == 2:
        sph = sph[:-1 - (f % 4)]
    cam = pwnfps_amd.spawn_camera((9, 4), ang_y=0.13 * f, ang_x=0.02 * f)
    return cam, 0.05 * f, sph


@pytest.mark.parametrize("slots", [2, 3, 4])
def test_frames_in_flight_are_the_oracles_frames(slots, oracle_lib):
    import pwnfps_amd
    from oracle import Oracle
    base = load_spheres("t0")
    r = pwnfps_amd.Renderer(W, H)
    r.level_load(level_path("pwnfps_level"))
    scale, pitch = 2, (W * 2 + 8) * 4                      # a padded surface pitch
    r.frames_config(slots, sbuf=True, zbuf=True, surface_scale=scale, pitch_bytes=pitch)
    O = Oracle()
    O.load_level(level_path("pwnfps_level"))
    nframes = 9
    got = {}
    for f in range(nframes + slots - 1):
        if f >= slots - 1:
            k = f - (slots - 1)
            fr = r.wait_frame(k % slots)
            assert fr["seq"] == k + 1
            got[k] = (fr["sbuf"].copy(), fr["zbuf"].copy(), fr["surface"].copy(), fr["sec"])
        if f < nframes:
            cam, sec, sph = _scene(f, base)
            r.set_objects(sph)                              # level_prepare_render, main.c:95: no wait
            if f >= 1 and slots >= 3:
                with pytest.raises(pwnfps_amd.PwnError) as e:
                    r.submit_frame(cam, sec, (f - 1) % slots)   # the previous frame's slot is still in flight
                assert e.value.code == -8                       # PWN_EBUSY
            r.submit_frame(cam, sec, f % slots)
    # depth is kept at pixels whose primary ray runs out of steps (trace.h:677); none does here,
    # so every plane is fully defined by its own frame
    for k in range(nframes):
        cam, sec, sph = _scene(k, base)
        O.set_spheres(sph)
        ob, oz = O.render(W, H, cam, sec=sec, blur=1)
        sb, zb, surf, fsec = got[k]
        assert fsec == np.float32(sec)
        assert (sb == ob).all(), "frame %d colour" % k
        assert (zb.view(np.uint32) == oz.view(np.uint32)).all(), "frame %d depth" % k
        # with a padded pitch the reference packs the rows its own way (screen.h:132,138-139);
        # bytes it never writes read 0 here
        up = O.upscale(ob, scale, pitch)
        assert surf.shape == up.shape and (surf == up).all(), "frame %d surface" % k
    # the blocking call can be mixed in and the slots re-configured once nothing is in flight
    cam, sec, sph = _scene(1, base)
    r.set_objects(sph)
    sb, zb = r.trace_screen_centred(cam, sec)
    assert (sb == got[1][0]).all()
    r.frames_config(2, sbuf=True)
    for _ in range(5):
        r.set_objects(sph)          # every copy of the tables is uploaded again: none may still be waiting for an event of the old slots
    r.submit_frame(cam, sec, 0)
    with pytest.raises(pwnfps_amd.PwnError):
        r.frames_config(3, sbuf=True)                       # a frame is in flight
    fr = r.wait_frame(0)
    assert (fr["sbuf"] == got[1][0]).all() and "zbuf" not in fr and "surface" not in fr
    r.frames_config(0)
    with pytest.raises(pwnfps_amd.PwnError):
        r.submit_frame(cam, sec, 0)
    r.close()


def test_frames_without_blur_and_at_4k(oracle_lib, cases):
    """POSTPROC_BLUR off (the trace writes the slot's plane directly) and the BASELINE frame
    size against the compiled reference's golden hashes, three frames in flight."""
    import pwnfps_amd
    want = [c for c in cases if c["name"] == "level_spawn_3840x2160"][0]
    r = pwnfps_amd.Renderer(3840, 2160)
    r.level_load(level_path("pwnfps_level"))
    r.set_objects(load_spheres("t0"))
    cam = np.array(want["cam"], np.float32)
    for blur, key in ((0, "pre"), (1, "post")):
        r.set_blur_passes(blur)
        r.frames_config(3, sbuf=True)
        for f in range(3):
            r.set_objects(load_spheres("t0"))
            r.submit_frame(cam, 0.0, f)
        for f in range(3):
            assert oracle_lib.fnv64(r.wait_frame(f)["sbuf"]) == want[key]
    r.close()


def test_trace_room_never_changes_a_frame(oracle_lib, cases):
    """PWN_OPT_TRACE_ROOM: the persistent trace grid leaves workgroups free for the other stream's kernels -- a fixed number,
    or (-1, the default) whatever the library's own comparison of 0 against one per CU says.  A frame is the golden frame
    whatever the setting, also while the setting changes under frames in flight; the state call reports what it did."""
    import pwnfps_amd
    c = [x for x in cases if x["name"] == "level_pose1_1280x720"][0]
    w, h = c["w"], c["h"]
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(level_path(c["level"]))
    r.set_objects(load_spheres(c["spheres"]))
    cam = np.array(c["cam"], np.float32)
    r.frames_config(3, sbuf=True)
    assert r.trace_room_state()["option"] == -1            # the library measures unless told otherwise
    for room, frames in ((0, 6), (256, 6), (64, 6), (4096, 4), (-1, 140)):
        r.set_trace_room(room)
        for f in range(frames + 2):
            if f >= 2:
                fr = r.wait_frame((f - 2) % 3)
                assert oracle_lib.fnv64(fr["sbuf"]) == c["post"], (room, f)
            if f < frames:
                r.submit_frame(cam, c["sec"], f % 3)
        st = r.trace_room_state()
        assert st["option"] == room
        if room >= 0:
            assert st["room_now"] == room and st["comparisons"] == 0
    # 140 delivered frames: windows of 24 with each setting (6 skipped after every change), then a choice
    assert st["comparisons"] >= 1 and st["changes"] >= 1 and st["room_now"] in (0, 256)        # (256 CUs: one workgroup per CU)
    with pytest.raises(pwnfps_amd.PwnError) as e:
        r.set_trace_room(-2)
    assert e.value.code == -1
    # the blocking call takes a fixed number too (and ignores the measuring mode: nothing runs beside it)
    r.set_trace_room(128)
    sb, _ = r.trace_screen_centred(cam, c["sec"])
    assert oracle_lib.fnv64(sb) == c["post"]
    r.close()


def test_a_launch_that_leaves_the_rotation_is_ordered_behind_the_launch_two_before_it(oracle_lib):
    """trace launches take their work-queue counters from sets used in turn and launch n resets the set of launch
    n - 2.  Frames in flight alternate between two streams, so launches two apart are on one stream; a blocking
    pwn_trace_screen_centred behind two frames in flight is launch n on the first stream with launch n - 2 on the SECOND: it has to
    wait for that frame's event (the wait was lost once in round 4 and nothing noticed: hence the count)."""
    import pwnfps_amd
    from oracle import Oracle
    base = load_spheres("t0")
    r = pwnfps_amd.Renderer(W, H)
    r.level_load(level_path("pwnfps_level"))
    r.frames_config(3, sbuf=True)
    O = Oracle()
    O.load_level(level_path("pwnfps_level"))
    assert r.launch_order_waits() == 0
    for f in range(5):                                      # frames 0 .. 4 on streams 0 1 0 1 0: the rotation itself needs no wait
        if f >= 3:
            r.wait_frame(f % 3)
        cam, sec, sph = _scene(f, base)
        r.set_objects(sph)
        r.submit_frame(cam, sec, f % 3)
    assert r.launch_order_waits() == 0
    # launch 5 by the blocking form: on the first stream, like launch 4; launch 3 went out on the second
    cam, sec, sph = _scene(5, base)
    r.set_objects(sph)
    sb = np.empty((H, W), np.uint32)
    r.trace_screen_centred(cam, sec, want_z=False, sbuf=sb)
    assert r.launch_order_waits() == 1
    O.set_spheres(sph)
    assert oracle_lib.fnv64(sb) == oracle_lib.fnv64(O.render(W, H, cam, sec=sec, blur=1)[0])
    for k in (3, 4):
        fr = r.wait_frame(k % 3)
        cam, sec, sph = _scene(k, base)
        O.set_spheres(sph)
        assert oracle_lib.fnv64(fr["sbuf"]) == oracle_lib.fnv64(O.render(W, H, cam, sec=sec, blur=1)[0])
    r.close()
