"""The refill scheduler of the trace kernel (trace_refill.hip: lanes whose ray ended take new
rays by ballot + prefix rank while the others walk on) renders the reference's pixels: golden
hashes at every BASELINE size, counters, odd frame sizes, every refill limit, strips."""
import os

import numpy as np
import pytest

from conftest import GOLD, level_path, load_spheres

pytestmark = pytest.mark.gpu


def _renderer(w, h, limit=None):
    import pwnfps_amd
    r = pwnfps_amd.Renderer(w, h)
    r.set_scheduler("refill")
    if limit is not None:
        r.set_refill_limit(limit)
    return r


def test_golden_cases_refill(oracle_lib, cases):
    by_size = {}
    for c in cases:
        by_size.setdefault((c["w"], c["h"]), []).append(c)
    for (w, h), cs in sorted(by_size.items()):
        for c in cs:
            r = _renderer(w, h)             # fresh context per case: depth starts at zero like the goldens
            r.level_load(level_path(c["level"]))
            r.set_objects(load_spheres(c["spheres"]))
            cam = np.array(c["cam"], np.float32)
            r.set_counters("steps" in c)
            post, z = r.trace_screen_centred(cam, c["sec"])
            st = r.stats()
            assert oracle_lib.fnv64(post) == c["post"], c["name"]
            assert oracle_lib.fnv64(z) == c["z"], c["name"]
            if "steps" in c:
                got = (st["rays"], st["steps"], st["portals"], st["sphere_tests"], st["exhausted"])
                want = (c["rays"], c["steps"], c["portals"], c["sphere_tests"], c["exhausted"])
                assert got == want, c["name"]
                assert st["wave_steps"] > 0 and st["phase_passes"] > 0
                assert st["phase_lanes"] == st["rays"]          # every ray is shaded in exactly one pass
            r.set_counters(False)
            r.set_blur_passes(0)
            pre, _ = r.trace_screen_centred(cam, c["sec"])
            assert oracle_lib.fnv64(pre) == c["pre"], c["name"]
            r.close()


@pytest.mark.parametrize("limit", [1, 7, 64, 200, 1000, 64000])
def test_every_refill_limit_gives_the_same_frame(limit, oracle_lib):
    """the limit only moves the point at which a wave shades its finished rays"""
    from oracle import Oracle
    import pwnfps_amd
    for lvl, key, size in (("pwnfps_level", "t0", (516, 270)), ("synth256", "synth256", (260, 131)), ("synth64", "synth64", (128, 96))):
        O = Oracle()
        O.load_level(level_path(lvl))
        sph = load_spheres(key)
        O.set_spheres(sph)
        data, _, spawn = O.get_level()
        w, h = size
        r = _renderer(w, h, limit)
        r.level_load(level_path(lvl))
        r.set_objects(sph)
        for ang in (0.0, 0.9, 2.4):
            cam = pwnfps_amd.spawn_camera(spawn, ang_y=ang, ang_x=0.1 * ang)
            blur = 1 if w % 4 == 0 else 0
            r.set_blur_passes(blur)
            a, za = r.trace_screen_centred(cam, 0.7)
            b, zb = O.render(w, h, cam, sec=0.7, blur=blur)
            assert (a == b).all(), (lvl, ang, int((a != b).sum()))
            # pixels whose primary ray runs out of steps keep the previous depth (trace.h:677)
            same = za.view(np.uint32) == zb.view(np.uint32)
            assert same.all() or lvl == "synth256", (lvl, ang)
        r.close()


def test_refill_strips_equal_full_frame(oracle_lib, cases):
    import torch
    c = next(x for x in cases if x["name"] == "level_pose1_1280x720")
    w, h = c["w"], c["h"]
    r = _renderer(w, h)
    r.level_load(level_path(c["level"]))
    r.set_objects(load_spheres(c["spheres"]))
    cam = np.array(c["cam"], np.float32)
    dev = torch.device("cuda:0")
    pre = torch.zeros((h, w), dtype=torch.int32, device=dev)
    z = torch.zeros((h, w), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    cuts = [0, 8, 13, 300, 301, 640, h]
    for y0, y1 in zip(cuts[:-1], cuts[1:]):
        r.trace_rows_device(cam, c["sec"], y0, y1, pre.data_ptr(), z.data_ptr(), stream)
    torch.cuda.synchronize()
    assert oracle_lib.fnv64(pre.cpu().numpy()) == c["pre"]
    assert oracle_lib.fnv64(z.cpu().numpy()) == c["z"]
    r.close()
