"""pwn_init_multi: ONE handle, one host loop (main.c:93-109), the frame row-tiled over several devices inside the
process (the reference's row parallelism is two OpenMP pragmas the host never sees, screen.h:63-67,77).  On a box with
one GPU the members are virtual -- the same ordinal N times, the in-process transport between them -- and everything
else is what N GPUs run: member threads, strips, halo rows, bounded blur with exact repeat, moving cuts, every member's own
copy of its strip into the caller's buffers.  Goldens of the compiled reference through the SAME calls a one-GPU host makes."""
import os

import numpy as np
import pytest

from conftest import GOLD, level_path, load_spheres

pytestmark = pytest.mark.gpu


def _group(w, h, n, device=0):
    import pwnfps_amd
    return pwnfps_amd.Renderer(w, h, devices=[device] * n)


def _case(cases, name):
    return [c for c in cases if c["name"] == name][0]


def _load(r, c):
    r.level_load(level_path(c["level"]))
    r.set_objects(load_spheres(c["spheres"]))
    return np.array(c["cam"], np.float32)


@pytest.mark.parametrize("members", [2, 3, 8])
@pytest.mark.parametrize("name", ["level_spawn_1280x720", "level_spawn_3840x2160", "synth256_cam0_1920x1080"])
def test_blocking_call_and_frames_in_flight_give_the_golden_frame(oracle_lib, cases, name, members):
    c = _case(cases, name)
    r = _group(c["w"], c["h"], members)
    gi = r.group_info()
    assert gi["members"] == members and gi["transport"] == "local" and gi["devices"] == [0] * members
    cam = _load(r, c)
    # ---- the blocking call (main.c:107), into the caller's own buffers
    sb = np.zeros((c["h"], c["w"]), np.uint32)
    zb = np.zeros((c["h"], c["w"]), np.float32)
    r.trace_screen_centred(cam, c["sec"], sbuf=sb, zbuf=zb)
    assert oracle_lib.fnv64(sb) == c["post"], (name, members)
    assert oracle_lib.fnv64(zb) == c["z"], (name, members)
    gi = r.group_info()
    assert gi["host_sink"] and gi["cuts"][0] == 0 and gi["cuts"][-1] == c["h"] and gi["frames"] == 1
    if name.startswith("synth256"):
        assert gi["frames_redone"] == 1 and gi["halo_rows"] == 0          # taps left the halo: repeated with whole strips, which stay
    else:
        assert gi["frames_redone"] == 0 and gi["halo_rows"] > 0
    r.close()
    # ---- frames in flight, delivered to the group's pinned frames (a fresh handle: this scene's depth carries over between frames)
    r = _group(c["w"], c["h"], members)
    cam = _load(r, c)
    r.frames_config(3, sbuf=True, zbuf=True)
    for i in range(3):
        r.set_objects(load_spheres(c["spheres"]))
        r.submit_frame(cam, c["sec"], i)
    f0 = r.wait_frame(0)
    assert oracle_lib.fnv64(f0["sbuf"]) == c["post"], (name, members)
    assert oracle_lib.fnv64(f0["zbuf"]) == c["z"], (name, members)
    assert f0["seq"] == 1
    exhausted = c.get("exhausted", 0) or 0
    for i in (1, 2):
        f = r.wait_frame(i)
        assert f["seq"] == i + 1
        if not exhausted:
            assert oracle_lib.fnv64(f["sbuf"]) == c["post"], (name, members, i)
    r.frames_config(0)
    r.close()


def test_frames_that_stay_on_the_devices_are_gathered_on_member_0(oracle_lib, cases):
    c = _case(cases, "level_spawn_1280x720")
    r = _group(c["w"], c["h"], 4)
    cam = _load(r, c)
    r.frames_config(3, sbuf=False)
    for i in range(5):
        if i >= 3:
            f = r.wait_frame(i % 3)
            assert f["d_sbuf"] and "sbuf" not in f
            assert oracle_lib.fnv64(r.read_plane(f["d_sbuf"])) == c["post"]
        r.set_objects(load_spheres(c["spheres"]))
        r.submit_frame(cam, c["sec"], i % 3)
    import time
    t0 = time.time()
    while not all(r.frame_ready(i) for i in (0, 1, 2)):          # pwn_frame_ready: without blocking, true before long
        assert time.time() - t0 < 10.0
    for i in (0, 1, 2):
        f = r.wait_frame(i)
        assert oracle_lib.fnv64(r.read_plane(f["d_sbuf"])) == c["post"]
        assert r.frame_ready(i)
    assert not r.group_info()["host_sink"]
    # the same handle, now with its frames delivered: the tiling is set up again behind the call
    r.frames_config(2, sbuf=True)
    r.submit_frame(cam, c["sec"], 0)
    assert oracle_lib.fnv64(r.wait_frame(0)["sbuf"]) == c["post"]
    assert r.group_info()["host_sink"]
    r.frames_config(0)
    r.close()


@pytest.mark.parametrize("members,scale,pad", [(2, 3, 0), (5, 2, 64), (3, 1, 0)])
def test_frames_in_flight_with_the_sdl_sink(oracle_lib, cases, members, scale, pad):
    """PWN_FRAME_SURFACE on a group: every member upscales its own strip (screen_upscale, screen.h:126-149) and copies those rows of the
    surface to the host; with a pitch wider than the rows the reference's packed layout, the bytes between rows 0"""
    c = _case(cases, "level_spawn_320x200")
    w, h = c["w"], c["h"]
    r = _group(w, h, members)
    cam = _load(r, c)
    pitch = w * scale * 4 + pad
    r.frames_config(3, sbuf=True, surface_scale=scale, pitch_bytes=pitch)
    O = oracle_lib.Oracle()
    for f in range(7):
        if f >= 3:
            fr = r.wait_frame(f % 3)
            assert oracle_lib.fnv64(fr["sbuf"]) == c["post"]
            want = O.upscale(fr["sbuf"], scale, pitch_bytes=pitch)
            assert fr["surface"].shape == want.shape and (fr["surface"] == want).all(), (f, members, scale, pad)
        if f < 4:
            r.set_objects(load_spheres(c["spheres"]))
            r.submit_frame(cam, c["sec"], f % 3)
    r.frames_config(0)
    # the same handle's blocking call behind it
    sb, _ = r.trace_screen_centred(cam, c["sec"])
    assert oracle_lib.fnv64(sb) == c["post"]
    r.close()


def test_one_object_table_behind_the_handle(oracle_lib):
    """obj_new / obj_set / obj_free (script.h:10-51) act on the handle's ONE table; level_prepare_render (main.c:95) brings every
    member's device up to date.  Against the oracle, frame by frame, while spheres move, appear and go."""
    w, h = 640, 400
    sph = load_spheres("t0")
    O = oracle_lib.Oracle()
    O.load_level(level_path("pwnfps_level"))
    r = _group(w, h, 3)
    r.level_load(level_path("pwnfps_level"))
    _, _, spawn = r.get_level()
    import pwnfps_amd
    cam = pwnfps_amd.spawn_camera(spawn, ang_y=0.3)
    ids = []
    for s in sph[:6]:
        o = r.obj_new()
        r.obj_set(o, "sphere", s["r"], s["refl"], s["x"], s["y"], s["z"], s["cb"], s["cg"], s["cr"])
        ids.append(o)
    live = list(range(6))
    for step in range(4):
        if step == 1:
            r.obj_free(ids[2]); live.remove(2)
        if step == 2:
            s = sph[7]
            o = r.obj_new()
            assert o == ids[2]                    # level_obj_new hands out the freed slot (level.h:41-62)
            r.obj_set(o, "sphere", s["r"], s["refl"], s["x"] + 0.25, s["y"], s["z"], s["cb"], s["cg"], s["cr"])
        r.level_prepare_render()
        got = r.get_objects()
        O.set_spheres(got)
        sb, zb = r.trace_screen_centred(cam, 0.25 * step)
        osb, ozb = O.render(w, h, cam, sec=0.25 * step, blur=1)
        assert (sb == osb).all() and (zb.view(np.uint32) == ozb.view(np.uint32)).all(), step
    counts, idx = r.get_bins()
    assert counts.sum() == len(idx)
    r.close()


def test_depth_carries_from_call_to_call_like_on_one_device(oracle_lib):
    """a pixel whose primary ray runs out of steps keeps the previous call's depth (trace.h:677): also when the rows are four devices'"""
    cams = np.load(os.path.join(GOLD, "levels", "synth256_cams.npy"))
    sph = load_spheres("synth256")
    w, h = 480, 272
    import pwnfps_amd
    out = []
    for members in (1, 4):
        r = pwnfps_amd.Renderer(w, h) if members == 1 else _group(w, h, members)
        r.level_load(level_path("synth256"))
        r.set_objects(sph)
        frames = [r.trace_screen_centred(cams[i % 3], 0.25 * i) for i in range(8)]
        out.append([(a.copy(), z.copy()) for a, z in frames])
        r.close()
    O = oracle_lib.Oracle()
    O.load_level(level_path("synth256"))
    O.set_spheres(sph)
    _, _, st = O.trace_rows(w, h, 0, h, cams[0])
    assert st.exhausted > 0
    for i, ((a, za), (b, zb)) in enumerate(zip(out[0], out[1])):
        assert (za.view(np.uint32) == zb.view(np.uint32)).all(), i
        assert (a == b).all(), i


def test_counters_add_up_over_the_members(cases):
    c = _case(cases, "level_spawn_1280x720")
    r = _group(c["w"], c["h"], 3)
    cam = _load(r, c)
    r.set_blur_passes(0)
    r.set_counters(True)
    pre = r.trace_screen_centred(cam, c["sec"], want_z=False)
    st = r.stats()
    assert (st["rays"], st["steps"], st["portals"], st["sphere_tests"], st["exhausted"]) == (c["rays"], c["steps"], c["portals"], c["sphere_tests"], c["exhausted"])
    r.set_counters(False)
    r.close()


def test_upscale_of_the_delivered_frame_and_what_a_group_refuses(oracle_lib, cases):
    import pwnfps_amd
    from pwnfps_amd import _lib
    c = _case(cases, "level_spawn_320x240")
    r = _group(c["w"], c["h"], 2)
    cam = _load(r, c)
    sb, _ = r.trace_screen_centred(cam, c["sec"])
    assert oracle_lib.fnv64(sb) == c["post"]
    O = oracle_lib.Oracle()
    big = r.screen_upscale(None, 3)                 # main.c:108 behind main.c:107
    assert (big == O.upscale(sb, 3)).all()
    for call in (lambda: r.tiled_info(), lambda: r.tiled_submit(cam, 0.0), lambda: r.tiled_wait(),
                 lambda: r.trace_rows_device(cam, 0.0, 0, 8, 1, 1)):
        with pytest.raises(pwnfps_amd.PwnError) as e:
            call()
        assert e.value.code == _lib.PWN_ENOTSUP
    # a registered host buffer, portable over the members' devices
    sb2 = np.zeros((c["h"], c["w"]), np.uint32)
    r.host_register(sb2)
    r.trace_screen_centred(cam, c["sec"], sbuf=sb2, want_z=False)
    assert (sb2 == sb).all()
    r.host_unregister(sb2)
    # the blocking call waits for nothing it did not start: frames in flight have to be collected first
    r.frames_config(2, sbuf=True)
    r.submit_frame(cam, c["sec"], 0)
    with pytest.raises(pwnfps_amd.PwnError) as e:
        r.trace_screen_centred(cam, c["sec"])
    assert e.value.code == _lib.PWN_EBUSY
    assert oracle_lib.fnv64(r.wait_frame(0)["sbuf"]) == c["post"]
    r.frames_config(0)
    r.close()
    # one device is pwn_init
    r1 = pwnfps_amd.Renderer(c["w"], c["h"], devices=[0])
    cam = _load(r1, c)
    assert oracle_lib.fnv64(r1.trace_screen_centred(cam, c["sec"])[0]) == c["post"]
    with pytest.raises(pwnfps_amd.PwnError):
        r1.group_info()
    r1.close()


def test_a_member_that_stops_answering_is_an_error_not_a_hang(oracle_lib, cases, monkeypatch):
    """the reference's error model is print-and-return (level.h:35-37,110-115); a group adds a failure the reference cannot have -- a
    device that does not answer -- and that comes back as PWN_ETIMEDOUT at the deadline (pwn_tiled_set_timeouts on the handle),
    naming the member; the next call sets the tiling up again and delivers the golden frame"""
    import time
    import pwnfps_amd
    from pwnfps_amd import _lib
    monkeypatch.setenv("PWN_DBG_GROUP_STALL", "2:2:2500")          # member 2 is 2.5 s late to the second blocking call
    c = _case(cases, "level_spawn_1280x720")
    r = _group(c["w"], c["h"], 3)
    cam = _load(r, c)
    r.tiled_set_timeouts(20.0, 0.5)
    sb, _ = r.trace_screen_centred(cam, c["sec"])
    assert oracle_lib.fnv64(sb) == c["post"]
    t0 = time.perf_counter()
    with pytest.raises(pwnfps_amd.PwnError) as e:
        r.trace_screen_centred(cam, c["sec"])
    dt = time.perf_counter() - t0
    assert e.value.code in (_lib.PWN_ETIMEDOUT, _lib.PWN_EHIP) and "member" in str(e.value), str(e.value)
    assert dt < 4.0, dt                                    # (the late member's thread is waited for when the tiling is taken down: 2.5 s, not for ever)
    sb, zb = r.trace_screen_centred(cam, c["sec"])
    assert oracle_lib.fnv64(sb) == c["post"] and oracle_lib.fnv64(zb) == c["z"]
    r.frames_config(2, sbuf=True)
    r.submit_frame(cam, c["sec"], 0)
    assert oracle_lib.fnv64(r.wait_frame(0)["sbuf"]) == c["post"]
    r.frames_config(0)
    r.close()


def test_preflight_report_stays_json_whatever_the_environment_holds(monkeypatch):
    """pwn_tiled_preflight pastes strings from the environment and the loader into its JSON text: a quote or a backslash in one of
    them (a dlopen error is the very case the report exists for) must not end the string early"""
    import pwnfps_amd
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", 'a"b\\c\td')
    for r in (pwnfps_amd.Renderer(64, 64), _group(64, 64, 2)):
        pre = r.tiled_preflight()                      # json.loads inside
        assert pre["HSA_ENABLE_IPC_MODE_LEGACY"] == "a?b?c?d" and pre["devices_visible"] >= 1
        r.close()


def test_rccl_that_does_not_come_up_is_replaced_by_copies(oracle_lib, cases):
    """RCCL between the members has never had to come up on this pool.  Where its bring-up fails -- here: two members on ONE device, which
    ncclCommInitRank refuses or the deadline ends -- the group goes on with copies between the members' planes, says so in
    pwn_group_info.note, and the frames are the goldens; a host that named the transport (PWN_GROUP_TRANSPORT=rccl) gets the error."""
    import pwnfps_amd
    c = _case(cases, "level_spawn_1280x720")
    os.environ["PWN_DBG_GROUP_TRY_RCCL"] = "1"
    try:
        r = pwnfps_amd.Renderer(c["w"], c["h"], devices=[0, 0])
        gi = r.group_info()
        if gi["transport"] != "rccl":
            r.close()
            pytest.skip("librccl did not load: " + gi["note"])
        r.tiled_set_timeouts(8, 8)
        cam = _load(r, c)
        sb, zb = r.trace_screen_centred(cam, c["sec"])
        assert oracle_lib.fnv64(sb) == c["post"] and oracle_lib.fnv64(zb) == c["z"]
        gi = r.group_info()
        assert gi["transport"] == "local" and "RCCL did not come up" in gi["note"], gi
        r.frames_config(3, sbuf=True)
        for i in range(3):
            r.submit_frame(cam, c["sec"], i)
        for i in range(3):
            assert oracle_lib.fnv64(r.wait_frame(i)["sbuf"]) == c["post"]
        r.frames_config(0)
        r.close()
        os.environ["PWN_GROUP_TRANSPORT"] = "rccl"
        r = pwnfps_amd.Renderer(c["w"], c["h"], devices=[0, 0])
        r.tiled_set_timeouts(8, 8)
        cam = _load(r, c)
        with pytest.raises(pwnfps_amd.PwnError):
            r.trace_screen_centred(cam, c["sec"])
        assert r.group_info()["transport"] == "rccl"
        r.close()
    finally:
        os.environ.pop("PWN_DBG_GROUP_TRY_RCCL", None)
        os.environ.pop("PWN_GROUP_TRANSPORT", None)


def test_two_devices_over_rccl(oracle_lib, cases):
    """the same handle on two real GPUs: one RCCL communicator rank per device, brought up inside the process"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import pwnfps_amd
    c = _case(cases, "level_spawn_3840x2160")
    r = pwnfps_amd.Renderer(c["w"], c["h"], devices=[0, 1])
    assert r.group_info()["transport"] == "rccl"
    cam = _load(r, c)
    sb, zb = r.trace_screen_centred(cam, c["sec"])
    assert oracle_lib.fnv64(sb) == c["post"] and oracle_lib.fnv64(zb) == c["z"]
    r.frames_config(3, sbuf=False)
    for i in range(3):
        r.submit_frame(cam, c["sec"], i)
    for i in range(3):
        assert oracle_lib.fnv64(r.read_plane(r.wait_frame(i)["d_sbuf"])) == c["post"]
    r.frames_config(0)
    r.close()
