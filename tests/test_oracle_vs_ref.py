"""Direct oracle <-> compiled-reference comparisons.  They run wherever
oracle/_ref/*.so exists (built by oracle/Makefile from /root/reference)."""
import os

import numpy as np
import pytest

import refharness
from conftest import GOLD, host_is_intel, level_path, load_spheres

pytestmark = pytest.mark.skipif(not refharness.available("tab"), reason="oracle/_ref not built")


def _pose(x, y, z, ay, ax):
    cy, sy = np.float32(np.cos(ay)), np.float32(np.sin(ay))
    cx, sx = np.float32(np.cos(ax)), np.float32(np.sin(ax))
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], np.float32)
    rx = np.array([[1, 0, 0], [0, cx, sx], [0, -sx, cx]], np.float32)
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = (rx @ ry).astype(np.float32)
    m[3, :3] = (x, y, z)
    return m


def test_random_scenes(oracle_lib):
    rng = np.random.default_rng(777)
    R = refharness.RefHarness("tab")
    for lvl in ("pwnfps_level", "synth64", "synth256"):
        R.load_level(level_path(lvl))
        O = oracle_lib.Oracle()
        O.load_level(level_path(lvl))
        data, _, _ = O.get_level()
        free = [(x, z) for z in range(64) for x in range(64) if chr(data[z, x]) in ';$"#&><,^']
        for it in range(12):
            x, z = free[rng.integers(len(free))]
            cam = _pose(x + rng.uniform(0.05, 0.95), rng.uniform(0.05, 0.95), z + rng.uniform(0.05, 0.95),
                        rng.uniform(0, 6.28), rng.uniform(-1.2, 1.2))
            sph = np.zeros(int(rng.integers(0, 20)), oracle_lib.SPHERE_DTYPE)
            for i in range(len(sph)):
                sph[i] = (rng.uniform(0.03, 0.4), rng.choice([0.0, 0.3, 0.6]), x + rng.uniform(-1, 2), rng.uniform(0.1, 1.2),
                          z + rng.uniform(-1, 2), *rng.uniform(0, 1.2, 3))
            sph["x"] = np.clip(sph["x"], 0.6, 62.4); sph["z"] = np.clip(sph["z"], 0.6, 62.4)
            sec = float(rng.uniform(0, 100))
            R.set_spheres(sph); O.set_spheres(sph)
            for blur in (0, 1):
                a, za = R.render(200, 152, cam, sec=sec, blur=blur)
                b, zb = O.render(200, 152, cam, sec=sec, blur=blur)
                assert (a == b).all(), (lvl, it, blur, int((a != b).sum()))
                assert (za.view(np.uint32) == zb.view(np.uint32)).all(), (lvl, it)


def test_w_lane_generality(oracle_lib):
    """v_dot / v_normalise are 4-lane (util.h:18-46): a camera matrix with
    non-zero w entries must still agree."""
    R = refharness.RefHarness("tab")
    R.load_level(level_path("pwnfps_level"))
    O = oracle_lib.Oracle()
    O.load_level(level_path("pwnfps_level"))
    sph = load_spheres("t0")
    R.set_spheres(sph); O.set_spheres(sph)
    cam = _pose(9.5, 0.5, 4.5, 0.3, 0.1)
    cam[0, 3], cam[1, 3], cam[2, 3], cam[3, 3] = 0.05, -0.02, 0.1, 0.7
    a, za = R.render(256, 128, cam, blur=1)
    b, zb = O.render(256, 128, cam, blur=1)
    assert (a == b).all() and (za.view(np.uint32) == zb.view(np.uint32)).all()


def test_native_and_table_builds_agree_on_intel():
    """The goldens come from the build whose rcpps / rsqrtps read captured tables; on an Intel host the untouched build
    (the host's own instructions) must render the same frames: one pose per level here, every golden case in
    tools/check_hw_goldens.py (frames.json `hw_equal`, asserted by test_every_golden_case_is_pinned_by_the_native_build)."""
    if not (host_is_intel() and refharness.available("hw")):
        pytest.skip("needs an Intel host")
    H = refharness.RefHarness("hw")
    T = refharness.RefHarness("tab")
    poses = (("pwnfps_level", "t0", _pose(9.5, 0.5, 4.5, 1.0, -0.1), 3.0, 320, 240),
             ("synth64", "synth64", np.load(os.path.join(GOLD, "levels", "synth64_cams.npy"))[2], 0.75, 384, 216),
             ("synth256", "synth256", np.load(os.path.join(GOLD, "levels", "synth256_cams.npy"))[1], 1.5, 384, 216))
    for lvl, sph, cam, sec, w, h in poses:
        for X in (H, T):
            X.load_level(level_path(lvl))
            X.set_spheres(load_spheres(sph))
        a, za = H.render(w, h, cam, sec=sec)
        b, zb = T.render(w, h, cam, sec=sec)
        assert (a == b).all() and (za.view(np.uint32) == zb.view(np.uint32)).all(), lvl


def test_every_golden_case_is_pinned_by_the_native_build(cases):
    """frames.json: every case's blurred frame and depth plane were reproduced by the reference built with the host's
    own rcpps / rsqrtps (tools/check_hw_goldens.py on the Intel build container) -- all of them, also the synthetic
    levels and the 8K frames, not only the small level.txt cases the generator checked."""
    missing = [c["name"] for c in cases if c.get("hw_equal") is not True]
    assert not missing, missing
    assert all("Intel" in c.get("hw_host", "") for c in cases)


def test_upscale_vs_reference(oracle_lib):
    R = refharness.RefHarness("tab")
    O = oracle_lib.Oracle()
    rng = np.random.default_rng(3)
    src = rng.integers(0, 2 ** 32, (9, 20), dtype=np.uint32)
    for scale, pitch in ((1, 80), (2, 176), (3, 240), (4, 336)):
        assert (R.upscale(src, scale, pitch) == O.upscale(src, scale, pitch)).all()


def test_edge_scenes(oracle_lib, tmp_path):
    """The oracle on the edge-of-domain scenes that the GPU parity test uses,
    against the reference's own code."""
    import edge_scenes
    R = refharness.RefHarness("tab")
    for sc in edge_scenes.scenes(oracle_lib.SPHERE_DTYPE):
        if not sc.ref_safe:
            continue
        O = oracle_lib.Oracle()
        if sc.text is None:
            path = level_path("pwnfps_level")
        else:
            path = str(tmp_path / (sc.name + ".txt"))
            with open(path, "w", newline="") as f:
                f.write(sc.text)
        R.load_level(path)
        O.load_level(path)
        R.set_spheres(sc.spheres)
        O.set_spheres(sc.spheres)
        blur = 1 if sc.w % 4 == 0 else 0
        a, za = R.render(sc.w, sc.h, sc.cam, sec=sc.sec, blur=blur)
        b, zb = O.render(sc.w, sc.h, sc.cam, sec=sc.sec, blur=blur)
        assert (a == b).all(), (sc.name, int((a != b).sum()))
        assert (za.view(np.uint32) == zb.view(np.uint32)).all(), sc.name


def test_random_levels_loader_three_way(oracle_lib, tmp_path):
    """level.txt format (level.h:107-228) on random files: the reference's own
    level_load, the oracle's and the product's (pwnfps_amd/csrc/level_host.c) parse
    to the same grid, portal table and spawn.  Letters stay off the grid border,
    where the reference's find_free_dir_2d reads outside lv->data (util.h:140-149)."""
    import ctypes as C
    import os
    from conftest import ROOT
    lib = C.CDLL(os.path.join(ROOT, "pwnfps_amd", "libpwnhip.so"))
    lib.pwn_parse_level.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(31337)
    R = refharness.RefHarness("tab")
    plain = list(';;;;;;$$##&&"<>,^....') + ['*']
    letters = [chr(c) for c in range(ord('A'), ord('Z') + 1)] + [chr(c) for c in range(ord('a'), ord('z') + 1)]
    for it in range(150):
        nrows, ncols = int(rng.integers(1, 66)), int(rng.integers(1, 70))
        eol = ["\n", "\r\n", "\r", "\n\n"][it % 4]
        rows = []
        for z in range(nrows):
            n = int(rng.integers(0, ncols + 1))
            row = [str(rng.choice(plain)) for _ in range(n)]
            if 1 <= z < min(nrows, 64) - 1:
                for x in range(1, min(n, 64) - 1):
                    if rng.random() < 0.04:
                        row[x] = str(rng.choice(letters))
            rows.append("".join(row))
        text = eol.join(rows) + (eol if it % 3 else "")
        path = str(tmp_path / ("lv%d.txt" % it))
        with open(path, "wb") as f:
            f.write(text.encode("latin-1"))
        R.load_level(path)
        d0, p0, s0 = R.get_level()
        O = oracle_lib.Oracle()
        O.load_level(path)
        d1, p1, s1 = O.get_level()
        cells = np.zeros(4096, np.uint8); pmap = np.zeros((26, 7), np.int32); spawn = np.zeros(2, np.int32)
        raw = text.encode("latin-1")
        assert lib.pwn_parse_level(raw, len(raw), cells.ctypes.data, pmap.ctypes.data, spawn.ctypes.data) == 0
        for name, (d, p, s) in (("oracle", (d1, p1, s1)), ("product", (cells.reshape(64, 64), pmap, spawn))):
            assert (np.asarray(d) == np.asarray(d0)).all(), (it, name, "grid")
            assert (np.asarray(p) == np.asarray(p0)).all(), (it, name, "pmap")
            assert (np.asarray(s) == np.asarray(s0)).all(), (it, name, "spawn")


def test_fuzz_scenes_oracle_vs_reference():
    """The scenes of the GPU fuzz campaign (tools/fuzz_parity.py, same generator and seed as
    tests/test_gpu_fuzz.py) with the compiled reference in place of the GPU: the oracle is
    pinned on exactly the inputs it is later the judge of."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "120", "3", "--ref"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert "120 scenes, 0 mismatches" in p.stdout


def test_axis_aligned_cameras_and_the_ramp_divide_by_zero(oracle_lib):
    """Axis-aligned cameras give rays with exactly rational slopes; where such a
    ray meets a ramp with ray.y == 0.5*ray.x the tilt divides by zero
    (trace.h:447-461).  Those pixels are the only ones at which the reference's
    two builds (shipped flags / plus -fno-finite-math-only) differ, and the
    oracle follows the second, i.e. plain SSE/IEEE behaviour."""
    if not refharness.available("nf"):
        pytest.skip("oracle/_ref/libpwnref_nf.so not built")
    R, N = refharness.RefHarness("tab"), refharness.RefHarness("nf")
    O = oracle_lib.Oracle()
    for X in (R, N, O):
        X.load_level(level_path("pwnfps_level"))
        X.set_spheres(load_spheres("t0"))
    seen = 0
    for (px, pz) in ((11.0, 6.0), (11.0, 5.5), (11.0, 5.0), (11.5, 4.5), (13.0, 2.5), (12.5, 6.0)):
        for (vx, vz) in ((0, -1), (0, 1), (1, 0), (-1, 0)):
            cam = np.zeros((4, 4), np.float32)
            cam[0], cam[1], cam[2], cam[3] = (vz, 0, -vx, 0), (0, 1, 0, 0), (vx, 0, vz, 0), (px, 0.5, pz, 1)
            a, za = R.render(320, 200, cam, sec=1.0, blur=0)
            n, zn = N.render(320, 200, cam, sec=1.0, blur=0)
            o, zo = O.render(320, 200, cam, sec=1.0, blur=0)
            assert (n == o).all() and (zn.view(np.uint32) == zo.view(np.uint32)).all(), (px, pz, vx, vz)
            assert (za.view(np.uint32) == zo.view(np.uint32)).all()
            differ = a != o
            assert not (differ & np.isfinite(zo)).any(), (px, pz, vx, vz)
            seen += int(differ.sum())
    assert seen > 0    # the case is really exercised


def test_far_start_scenes(oracle_lib):
    """tests/golden/far_starts.npz: lattice scenes of tools/fuzz_parity.py (seeds 9002, 9004) in which a bounced
    ray starts its next segment far outside the grid (its cell number beyond 16 bits).  The oracle against the
    compiled reference on them; tests/test_gpu_parity.py holds the GPU against the oracle on the same scenes."""
    import os
    from conftest import GOLD
    k = np.load(os.path.join(GOLD, "far_starts.npz"))
    R = refharness.RefHarness("nf" if refharness.available("nf") else "tab")
    O = oracle_lib.Oracle()
    for i in range(len(k["names"])):
        text, cam, sph, sec = str(k["text_%d" % i]), k["cam_%d" % i], k["sph_%d" % i], float(k["sec_%d" % i])
        w, h = (int(v) for v in k["wh_%d" % i])
        import tempfile
        with tempfile.NamedTemporaryFile("wb", suffix=".txt", delete=False) as f:
            f.write(text.encode("latin-1"))
        R.load_level(f.name)
        os.unlink(f.name)
        O.load_level_text(text)
        R.set_spheres(sph); O.set_spheres(sph)
        a, za = R.render(w, h, cam, sec=sec, blur=0)
        b, zb = O.render(w, h, cam, sec=sec, blur=0)
        assert (za.view(np.uint32) == zb.view(np.uint32)).all(), k["names"][i]
        assert (a == b).all(), (k["names"][i], int((a != b).sum()))

