"""The oracle (oracle/pwn_oracle.c) against the golden vectors produced by
the compiled reference (tools/gen_goldens.py).  CPU only, sizes kept small."""
import os

import numpy as np
import pytest

from conftest import GOLD, level_path, load_spheres

LEVELS = ["pwnfps_level", "synth64", "synth256"]


@pytest.mark.parametrize("name", LEVELS)
def test_level_loader_and_bins(oracle_lib, name):
    t = np.load(os.path.join(GOLD, "levels", name + "_tables.npz"))
    O = oracle_lib.Oracle()
    O.load_level(level_path(name))
    data, pmap, spawn = O.get_level()
    assert (data == t["data"]).all()
    assert (pmap == t["pmap"]).all()
    assert (spawn == t["spawn"]).all()
    key = "t0" if name == "pwnfps_level" else name
    O.set_spheres(load_spheres(key))
    counts, idx = O.get_bins()
    assert (counts == t["bin_counts"]).all()
    assert (idx == t["bin_idx"]).all()


def test_level_txt_known_tables(oracle_lib):
    """spawn, the lower-case quirk, rotations."""
    O = oracle_lib.Oracle()
    O.load_level(level_path("pwnfps_level"))
    data, pmap, spawn = O.get_level()
    assert tuple(spawn) == (9, 4)
    assert chr(data[4, 4]) == "N" and chr(data[24, 13]) == "Y"   # 'm' -> N, 'x' -> Y
    A, E, M, N, Y = (pmap[ord(c) - 65] for c in "AEMNY")
    assert tuple(A[:5]) == (13, 7, 7, 8, 0)
    assert tuple(E[:5]) == (22, 5, 15, 19, 1)
    assert tuple(M[:4]) == (4, 3, 4, 4)
    assert tuple(N[:4]) == (4, 4, -1, -1)            # unpaired
    assert tuple(Y[:5]) == (3, 12, 13, 24, 1)
    assert (pmap[[20, 21, 22], 0] == -1).all()        # U, V, W unused


def _small(c):
    return c["w"] * c["h"] <= 1280 * 720


def test_frames_vs_golden(oracle_lib, cases):
    ran = 0
    for c in cases:
        if not _small(c):
            continue
        O = oracle_lib.Oracle()
        O.load_level(level_path(c["level"]))
        O.set_spheres(load_spheres(c["spheres"]))
        cam = np.array(c["cam"], np.float32)
        pre, z, st = O.render(c["w"], c["h"], cam, sec=c["sec"], blur=0, stats=True)
        post, z2 = O.render(c["w"], c["h"], cam, sec=c["sec"], blur=1)
        assert oracle_lib.fnv64(pre) == c["pre"], c["name"]
        assert oracle_lib.fnv64(post) == c["post"], c["name"]
        assert oracle_lib.fnv64(z) == c["z"], c["name"]
        assert oracle_lib.fnv64(z2) == c["z"], c["name"]
        if "steps" in c:
            got = (st.rays, st.steps, st.portals, st.sphere_tests, st.exhausted)
            want = (c["rays"], c["steps"], c["portals"], c["sphere_tests"], c["exhausted"])
            assert got == want, c["name"]
        ran += 1
    assert ran >= 15


def test_some_case_exhausts_maxsteps(cases):
    assert any(c.get("exhausted", 0) > 0 for c in cases), "no golden exercises trace.h:677"


def test_raw_frames(oracle_lib):
    raw = np.load(os.path.join(GOLD, "raw_320x240.npz"))
    O = oracle_lib.Oracle()
    O.load_level(level_path("pwnfps_level"))
    O.set_spheres(load_spheres("t0"))
    cam = np.eye(4, dtype=np.float32)
    cam[3, :3] = (9.5, 0.5, 4.5)
    pre, z = O.render(320, 240, cam, blur=0)
    post, _ = O.render(320, 240, cam, blur=1)
    assert (pre == raw["pre"]).all()
    assert (post == raw["post"]).all()
    assert (z.view(np.uint32) == raw["z"].view(np.uint32)).all()


def test_strip_of_1080p(oracle_lib):
    """Rows [512,544) of the synth64 cam1 1080p frame: the strip form must give
    the same rows as the full frame (pixels are independent; blur rows are
    seeded per row, screen.h:82)."""
    s = np.load(os.path.join(GOLD, "strips.npz"))
    y0, y1 = (int(v) for v in s["c3_rows"])
    O = oracle_lib.Oracle()
    O.load_level(level_path("synth64"))
    O.set_spheres(load_spheres("synth64"))
    cam = np.load(os.path.join(GOLD, "levels", "synth64_cams.npy"))[1]
    pre, z, _ = O.trace_rows(1920, 1080, y0, y1, cam, sec=0.25)
    assert (pre[y0:y1] == s["c3_pre"]).all()
    assert (z[y0:y1].view(np.uint32) == s["c3_z"].view(np.uint32)).all()
    assert not pre[:y0].any() and not pre[y1:].any()      # other rows untouched


def test_campaign(oracle_lib):
    c = np.load(os.path.join(GOLD, "campaign.npz"))
    for i in range(len(c["cams"])):
        O = oracle_lib.Oracle()
        O.load_level(level_path(str(c["level"][i])))
        ns = int(c["hashes"][i][3])
        O.set_spheres(c["spheres"][i][:ns])
        w, h = (int(v) for v in c["size"][i])
        pre, z = O.render(w, h, c["cams"][i], sec=float(c["sec"][i]), blur=0)
        post, _ = O.render(w, h, c["cams"][i], sec=float(c["sec"][i]), blur=1)
        assert [oracle_lib.fnv64(pre), oracle_lib.fnv64(post), oracle_lib.fnv64(z)] == list(c["hashes"][i][:3]), i


def test_thread_count_invariance(oracle_lib):
    O = oracle_lib.Oracle()
    O.load_level(level_path("pwnfps_level"))
    O.set_spheres(load_spheres("t0"))
    cam = np.eye(4, dtype=np.float32)
    cam[3, :3] = (9.5, 0.5, 4.5)
    a, za = O.render(320, 240, cam, threads=1)
    b, zb = O.render(320, 240, cam, threads=4)
    assert (a == b).all() and (za.view(np.uint32) == zb.view(np.uint32)).all()


def test_step_map_adds_up(oracle_lib):
    """pwno_step_map (the analysis aid behind tools/unit_shapes.py): the per-pixel, per-segment walk iterations
    add up to the frame's step counter, every pixel has a primary segment, and the 16x4 units of the level.txt
    frame at 4K cost what the GPU kernel's own wave-iteration counter reports for that shape (1 741 949: checked
    in the tool's output, too long for this suite) -- here a 320x240 frame keeps the bookkeeping honest."""
    import ctypes as C
    O = oracle_lib.Oracle()
    O.load_level(level_path("pwnfps_level"))
    O.set_spheres(load_spheres("t0"))
    cam = np.eye(4, dtype=np.float32)
    cam[3, :3] = (9.5, 0.5, 4.5)
    L = oracle_lib.lib()
    L.pwno_step_map.argtypes = [C.c_void_p]
    L.pwno_step_map.restype = None
    m = np.zeros((240, 320, 3), np.uint16)
    L.pwno_step_map(m.ctypes.data)
    try:
        sb, zb, st = O.trace_rows(320, 240, 0, 240, cam)
    finally:
        L.pwno_step_map(None)
    assert int(m.sum(dtype=np.int64)) == st.steps
    assert (m[:, :, 0] > 0).all()
    assert int((m > 0).sum()) == st.rays
    sb2, _, st2 = O.trace_rows(320, 240, 0, 240, cam)            # switched off again: same frame, nothing written
    assert (sb == sb2).all() and st2.steps == st.steps


def test_loader_edge_cases(oracle_lib):
    O = oracle_lib.Oracle()
    # empty text: all wall, spawn 0,0 (level_new, level.h:85-105)
    O.load_level_text("")
    d, p, s = O.get_level()
    assert (d == ord(".")).all() and (p[:, 0] == -1).all() and tuple(s) == (0, 0)
    # CRLF, blank lines vanish, ragged rows, EOF without newline
    O.load_level_text("..;\r\n\r\n\n.*\r\n;;;;")
    d, p, s = O.get_level()
    assert bytes(d[0, :4]) == b"..;." and bytes(d[1, :3]) == b".;." and bytes(d[2, :5]) == b";;;;."
    assert tuple(s) == (1, 1)
    # a 64-column row needs no newline; the newline that follows is swallowed
    O.load_level_text(";" * 64 + "\n" + "#" * 3 + "\n")
    d, _, _ = O.get_level()
    assert (d[0] == ord(";")).all() and bytes(d[1, :4]) == b"###."
    # more than 64 rows: the rest is ignored
    O.load_level_text(("$\n" * 70))
    d, _, _ = O.get_level()
    assert (d[:, 0] == ord("$")).all() and (d[:, 1] == ord(".")).all()
    # lower-case quirk (level.h:144-161): 'a' registers in A, is stored as 'B'
    # and registers in B too; 'z' is not a portal
    O.load_level_text(".....\n.;a;.\n.;A;.\n..z..\n")
    d, p, _ = O.get_level()
    assert chr(d[1, 2]) == "B" and chr(d[3, 2]) == "z"
    assert tuple(p[0][:4]) == (2, 1, 2, 2) and tuple(p[1][:4]) == (2, 1, -1, -1)
    # third sighting of a letter is ignored (level.h:149-160)
    O.load_level_text(".;C;C;C;.\n")
    _, p, _ = O.get_level()
    assert tuple(p[2][:4]) == (2, 0, 4, 0)


def test_sphere_bins_edge_cases(oracle_lib):
    from oracle import SPHERE_DTYPE
    O = oracle_lib.Oracle()
    O.load_level(level_path("pwnfps_level"))
    s = np.zeros(3, SPHERE_DTYPE)
    s[0] = (0.3, 0.5, 9.5, 0.3, 5.5, 1, 1, 1)          # one cell
    s[1] = (0.6, 0.5, 10.0, 0.3, 6.0, 1, 1, 1)         # 2x2 cells: (9..10, 5..6)
    s[2] = (0.5, 0.5, 0.2, 0.3, 63.8, 1, 1, 1)         # pokes outside the grid: clipped
    O.set_spheres(s)
    counts, idx = O.get_bins()
    c = counts.reshape(64, 64)
    assert c[5, 9] == 2 and c[5, 10] == 1 and c[6, 9] == 1 and c[6, 10] == 1
    assert c[63, 0] == 1 and counts.sum() == 1 + 4 + 1
    # list order inside a cell is object order (level.h:76-79)
    off = np.concatenate([[0], np.cumsum(counts.astype(np.int64))]).astype(int)
    cell = 5 * 64 + 9
    assert list(idx[off[cell]:off[cell + 1]]) == [0, 1]
    O.set_spheres(s[:0])
    counts, idx = O.get_bins()
    assert counts.sum() == 0 and len(idx) == 0
