"""The plain-C host (host/pwnhost.c) over the C ABI: the reference's frame loop
(main.c:93-109) with libpwnhip.so in place of trace_screen_centred /
screen_upscale.  CPU: it builds with gcc alone and reports errors like the
ABI says.  GPU: its frame is the compiled reference's golden frame."""
import os
import re
import subprocess
import time

import numpy as np
import pytest

from conftest import GOLD, ROOT, level_path

HOST = os.path.join(ROOT, "host", "pwnhost")


@pytest.fixture(scope="module")
def host_bin():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "pwnfps_amd", "csrc")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)
    return HOST


def test_host_builds_with_gcc_only_and_links_the_abi(host_bin):
    out = subprocess.check_output(["nm", "-D", "--undefined-only", host_bin]).decode()
    used = set(re.findall(r"U (pwn_\w+)", out))
    assert {"pwn_init", "pwn_level_load", "pwn_upload_spheres", "pwn_trace_screen_centred",
            "pwn_screen_upscale", "pwn_destroy"} <= used
    # nothing but the C ABI, libc and libm: no HIP / C++ symbols on the host side
    assert not re.search(r"U (hip|_Z)", out)


def test_host_usage_and_bad_option(host_bin):
    p = subprocess.run([host_bin], capture_output=True)
    assert p.returncode == 2 and b"usage" in p.stderr
    p = subprocess.run([host_bin, level_path("pwnfps_level"), "-Q", "1"], capture_output=True)
    assert p.returncode == 2 and b"unknown option" in p.stderr
    p = subprocess.run([host_bin, level_path("pwnfps_level"), "-q", "1"], capture_output=True)
    assert p.returncode == 2 and b"-q takes 2.." in p.stderr


def test_lua_leg_is_conditional(host_bin):
    """host/lua_host.c (script.h:71-103 over the ABI's object calls) is built in where Lua 5.1 is found;
    elsewhere -l says so instead of silently doing something else."""
    out = subprocess.check_output(["make", "-s", "-C", os.path.join(ROOT, "host"), "lua-leg"]).decode()
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", host_bin]).decode()
    if "not found" in out:
        assert "lua_pcall" not in syms
        p = subprocess.run([host_bin, level_path("pwnfps_level"), "-l", "game.lua"], capture_output=True)
        assert p.returncode == 2 and b"built without Lua" in p.stderr
    else:
        assert "lua_pcall" in syms and "luaL_loadfile" in syms
    src = open(os.path.join(ROOT, "host", "lua_host.c")).read()
    for name in ("obj_new", "obj_set", "obj_free", "level_get", "level_set", "on_tick", "luaL_openlibs", "luaL_loadfile"):
        assert name in src


@pytest.mark.gpu
def test_host_frame_is_the_reference_frame(host_bin, cases, tmp_path, oracle_lib):
    # the reference's shipped configuration: 320x200, x3 upscale (defs.h:11-15)
    want = [c for c in cases if c["name"] == "level_spawn_320x200"][0]
    ppm = str(tmp_path / "f.ppm")
    p = subprocess.run([host_bin, level_path("pwnfps_level"), "-s", os.path.join(GOLD, "spheres_t0.txt"),
                        "-w", "320", "-h", "200", "-x", "3", "-o", ppm], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    out = p.stdout.decode()
    assert "spawn: 9 4" in out
    m = re.search(r"sbuf fnv64 ([0-9a-f]{16}), surface fnv64 ([0-9a-f]{16})", out)
    assert m and m.group(1) == want["post"]
    # the surface is the oracle's screen_upscale of that frame, and the PPM holds it
    from oracle import Oracle
    O = Oracle()
    O.load_level(level_path("pwnfps_level"))
    O.set_spheres(np.load(os.path.join(GOLD, "spheres_t0.npy")))
    sb, _ = O.render(320, 200, np.array(want["cam"], np.float32), sec=0.0, blur=1)
    up = O.upscale(sb, 3)
    assert oracle_lib.fnv64(up) == m.group(2)
    raw = open(ppm, "rb").read()
    assert raw.startswith(b"P6\n960 600\n255\n")
    rgb = np.frombuffer(raw[len(b"P6\n960 600\n255\n"):], np.uint8).reshape(600, 960, 3)
    assert (rgb[..., 0] == ((up >> 16) & 255)).all() and (rgb[..., 2] == (up & 255)).all()


@pytest.mark.gpu
def test_one_process_several_gpus_shows_the_same_frames(host_bin):
    """-G N: the SAME process and the same loop (main.c:93-109), its one handle made by pwn_init_multi -- here N members on the one
    device of -d 0.  Frame by frame what the one-GPU loop shows: scripted spheres through the handle's one object table, a
    turning camera, depth carried from call to call; blocking and with three frames in flight."""
    args = [host_bin, level_path("pwnfps_level"), "-g", os.path.join(ROOT, "pwnfps_amd", "data", "game_objects.txt"),
            "-w", "640", "-h", "360", "-x", "2", "-n", "12", "-t", "0.05", "-a", "0.07", "-v", "1"]
    a = subprocess.run(args, capture_output=True, timeout=300)
    assert a.returncode == 0, a.stderr.decode()
    fa = re.findall(r"frame (\d+) sec (\S+) fnv64 ([0-9a-f]{16})", a.stdout.decode())
    sa = re.search(r"surface fnv64 ([0-9a-f]{16})", a.stdout.decode()).group(1)
    assert len(fa) == 12
    for extra in (["-G", "3", "-d", "0"], ["-G", "2", "-d", "0", "-q", "3"]):
        b = subprocess.run(args + extra, capture_output=True, timeout=300)
        assert b.returncode == 0, (extra, b.stderr.decode())
        out = b.stdout.decode()
        assert re.findall(r"frame (\d+) sec (\S+) fnv64 ([0-9a-f]{16})", out) == fa, extra
        assert re.search(r"surface fnv64 ([0-9a-f]{16})", out).group(1) == sa, extra
        if "-q" not in extra:
            assert "group of 3 in one process" in out and "12 frames" in out
    # -G goes with one process
    c = subprocess.run(args + ["-G", "2", "-W", "2"], capture_output=True, timeout=60)
    assert c.returncode == 2


@pytest.mark.gpu
def test_host_frames_in_flight_present_the_same_frames(host_bin):
    """-q 3: the loop of main.c:93-140 with three frames in flight (pinned sbuf + surface through
    pwn_submit_frame / pwn_wait_frame) shows, frame by frame, what the blocking loop shows:
    scripted spheres (changed between submits without waiting), a turning camera, x2 upscale."""
    args = [host_bin, level_path("pwnfps_level"), "-g", os.path.join(ROOT, "pwnfps_amd", "data", "game_objects.txt"),
            "-w", "640", "-h", "360", "-x", "2", "-n", "12", "-t", "0.05", "-a", "0.07", "-v", "1"]
    a = subprocess.run(args, capture_output=True, timeout=300)
    b = subprocess.run(args + ["-q", "3"], capture_output=True, timeout=300)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr.decode(), b.stderr.decode())
    fa = re.findall(r"frame (\d+) sec (\S+) fnv64 ([0-9a-f]{16})", a.stdout.decode())
    fb = re.findall(r"frame (\d+) sec (\S+) fnv64 ([0-9a-f]{16})", b.stdout.decode())
    assert len(fa) == 12 and fa == fb
    assert len({h for _, _, h in fa}) == 12                  # the frames do differ from each other
    sa = re.search(r"surface fnv64 ([0-9a-f]{16})", a.stdout.decode()).group(1)
    sb = re.search(r"surface fnv64 ([0-9a-f]{16})", b.stdout.decode()).group(1)
    assert sa == sb


@pytest.mark.gpu
def test_host_row_tiled_over_three_processes(host_bin, tmp_path):
    """host/pwnhost -W 3: three processes of the plain-C host, one per rank (all on the one GPU here, over
    the shared-memory transport), present the frames of the blocking single-process loop: scripted
    spheres, a turning camera."""
    base = [host_bin, level_path("pwnfps_level"), "-g", os.path.join(ROOT, "pwnfps_amd", "data", "game_objects.txt"),
            "-w", "640", "-h", "360", "-x", "1", "-n", "8", "-t", "0.05", "-a", "0.07", "-v", "1"]
    a = subprocess.run(base, capture_output=True, timeout=300)
    assert a.returncode == 0, a.stderr.decode()
    fa = re.findall(r"frame (\d+) sec \S+ fnv64 ([0-9a-f]{16})", a.stdout.decode())
    idfile = str(tmp_path / "group.id")
    # an id file left behind by an earlier run that died: the ranks must wait for THIS launch's id, not take that one
    with open(idfile, "wb") as f:
        f.write(b"pwnid \n" + b"/pwn_tiled_stale_0" + bytes(110))
    os.utime(idfile, (time.time() - 3600, time.time() - 3600))
    procs = [subprocess.Popen(base + ["-W", "3", "-R", str(r), "-I", idfile, "-T", "shm"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(3)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e.decode()
        outs.append(o.decode())
    fb = re.findall(r"frame (\d+) sec \S+ fnv64 ([0-9a-f]{16})", outs[0])
    assert len(fa) == 8 and fa == fb
    # (the cuts move with what the strips cost: the rows a rank ends with are not the equal split's)
    assert re.search(r"rank 0 of 3: rows \[0,\d+\) now \(the cuts moved \d+ times; 120 rows to begin with\), halo 19 rows, 8 frames \(0 repeated", outs[0])
    assert re.search(r"rank 2 of 3: rows \[\d+,360\) now", outs[2])
    sa = re.search(r"surface fnv64 ([0-9a-f]{16})", a.stdout.decode()).group(1)
    assert re.search(r"surface fnv64 ([0-9a-f]{16})", outs[0]).group(1) == sa
    # -M 1: every rank copies its strip into one frame in POSIX shared memory (pwn_tiled_host_sink); every rank
    # then sees every whole frame
    # (the same id file again, right after the first launch, with a launch nonce: -N)
    with open(idfile, "wb") as f:
        f.write(b"pwnid launch1\n" + b"/pwn_tiled_stale_1" + bytes(110))
    procs = [subprocess.Popen(base + ["-W", "3", "-R", str(r), "-I", idfile, "-N", "launch2", "-T", "shm", "-M", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(3)]
    for r, p in enumerate(procs):
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e.decode()
        assert re.findall(r"frame (\d+) sec \S+ fnv64 ([0-9a-f]{16})", o.decode()) == fa, r
        if r == 0:
            assert re.search(r"surface fnv64 ([0-9a-f]{16})", o.decode()).group(1) == sa


    # -O 1: the gather's root rotates (pwn_tiled_gather_root): frame f is presented by rank f mod 3, the last one (7) by rank 1
    procs = [subprocess.Popen(base + ["-W", "3", "-R", str(r), "-I", idfile, "-N", "launch3", "-T", "shm", "-O", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(3)]
    for r, p in enumerate(procs):
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e.decode()
        assert re.findall(r"frame (\d+) sec \S+ fnv64 ([0-9a-f]{16})", o.decode()) == [x for x in fa if int(x[0]) % 3 == r], r
        assert (re.search(r"surface fnv64 ([0-9a-f]{16})", o.decode()) is not None) == (r == 1)
        if r == 1:
            assert re.search(r"surface fnv64 ([0-9a-f]{16})", o.decode()).group(1) == sa

    # -W 3 alone: the host forks the ranks itself (all on device 0 here)
    p = subprocess.run(base[:-2] + ["-W", "3", "-d", "0", "-T", "shm", "-M", "1"], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    o = p.stdout.decode()
    assert re.search(r"surface fnv64 ([0-9a-f]{16})", o).group(1) == sa
    assert all(("rank %d of 3" % r) in o for r in range(3))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["blocking", "in_flight"])
def test_host_interactive_walk_every_frame_is_the_oracles(host_bin, mode, oracle_lib):
    """pwnhost -k: the interactive loop without a window (main.c:93-379): a key script walks and turns the
    player through the level (through a portal too), the game script moves the spheres, the frame
    goes through the x2 sink.  Every frame's hash is the oracle's for the camera that host/player.c --
    driven here from Python over the same key script -- has at that frame."""
    import ctypes as C
    import test_player as tp
    keys = os.path.join(GOLD, "keys_walk.txt")
    n, dt, w, h = 160, 1.0 / 30.0, 320, 200
    args = [host_bin, level_path("pwnfps_level"), "-g", os.path.join(ROOT, "pwnfps_amd", "data", "game_objects.txt"),
            "-k", keys, "-w", str(w), "-h", str(h), "-x", "2", "-n", str(n), "-t", repr(dt), "-v", "1"]
    if mode == "in_flight":
        args += ["-q", "3"]
    p = subprocess.run(args, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()
    got = re.findall(r"frame (\d+) sec \S+ fnv64 ([0-9a-f]{16})", p.stdout.decode())
    assert len(got) == n
    # the same run on the CPU: player library + restated script + oracle
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), os.path.join(ROOT, "host", "libpwnplayer.so")])
    L = C.CDLL(os.path.join(ROOT, "host", "libpwnplayer.so"))
    L.pwn_player_init.argtypes = [C.POINTER(tp.Player), C.c_void_p]
    L.pwn_player_step.argtypes = [C.POINTER(tp.Player), C.POINTER(tp.Keys), C.c_float, C.c_void_p, C.c_void_p]
    L.pwn_keys_event.argtypes = [C.POINTER(tp.Keys), C.c_int, C.c_int]
    L.pwn_keys_load.argtypes = [C.c_char_p, C.POINTER(tp.KeyEvent), C.c_int]
    ev = (tp.KeyEvent * 64)()
    nev = L.pwn_keys_load(keys.encode(), ev, 64)
    assert nev == 10
    t = np.load(os.path.join(GOLD, "levels", "pwnfps_level_tables.npz"))
    data, pmap, spawn = (np.ascontiguousarray(t[k]) for k in ("data", "pmap", "spawn"))
    from oracle import Oracle
    from pwnfps_amd.script import GameScript, ObjectTable, frame_times
    O = Oracle()
    O.load_level(level_path("pwnfps_level"))
    T = ObjectTable(data)
    g = GameScript(T)
    secs, ticks = frame_times(n, dt)
    pl, kk = tp.Player(), tp.Keys()
    L.pwn_player_init(C.byref(pl), spawn.astype(np.int32).ctypes.data)
    traversed = 0
    for f in range(n):
        O.set_spheres(T.live())
        img, _ = O.render(w, h, np.array(list(pl.cam), np.float32).reshape(4, 4), sec=secs[f], blur=1)
        assert got[f] == (str(f), oracle_lib.fnv64(img)), (mode, f)
        g.on_tick(*ticks[f])
        for e in ev[:nev]:
            if e.frame == f:
                L.pwn_keys_event(C.byref(kk), e.sym, e.down)
        L.pwn_player_step(C.byref(pl), C.byref(kk), np.float32(dt), data.astype(np.uint8).ctypes.data, pmap.astype(np.int32).ctypes.data)
        traversed = pl.traversals
    assert len({h_ for _, h_ in got}) > 100                 # the camera does move
    assert traversed >= 1, "the walk was laid out to go through a portal"


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["shm", "rccl"])
def test_host_with_a_rank_that_never_comes_prints_and_returns(host_bin, transport, tmp_path):
    """pwnhost -W 2 with only rank 0 started: the tiling's bring-up gives up after -X seconds (pwn_tiled_set_timeouts), the host
    prints the library's message -- which rank, waiting for what -- and leaves with status 2.  print-and-return, like
    level.h:35-37,110-115; never a hang."""
    import time
    t0 = time.time()
    p = subprocess.run([host_bin, level_path("pwnfps_level"), "-w", "640", "-h", "360", "-n", "3", "-W", "2", "-R", "0",
                        "-I", str(tmp_path / "id"), "-T", transport, "-X", "3", "-Y", "2"], capture_output=True, timeout=120)
    assert p.returncode == 2, (p.returncode, p.stderr[-2000:])
    err = p.stderr.decode()
    assert "deadline" in err and "rank 0" in err, err
    assert time.time() - t0 < 60


@pytest.mark.gpu
def test_host_missing_level_reports_eio(host_bin):
    p = subprocess.run([host_bin, "/nonexistent/level.txt"], capture_output=True, timeout=120)
    assert p.returncode == 1 and b"level file could not be read" in p.stderr
