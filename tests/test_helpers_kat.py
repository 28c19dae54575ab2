"""Known-answer tests of the helper routines (util.h) and libm, oracle side."""
import os

import numpy as np

from conftest import GOLD


def test_libm_kat(oracle_lib):
    k = np.load(os.path.join(GOLD, "libm_kat.npz"))
    L = oracle_lib.lib()
    x = k["x_sincos"]
    s = np.array([L.pwno_sinf(float(v)) for v in x], np.float32)
    c = np.array([L.pwno_cosf(float(v)) for v in x], np.float32)
    assert (s.view(np.uint32) == k["sinf"].view(np.uint32)).all()
    assert (c.view(np.uint32) == k["cosf"].view(np.uint32)).all()
    xe = k["x_exp"]
    e = np.array([L.pwno_expf(float(v)) for v in xe], np.float32)
    assert (e.view(np.uint32) == k["expf"].view(np.uint32)).all()


def test_col_ftoint(oracle_lib):
    k = np.load(os.path.join(GOLD, "helpers_kat.npz"))
    L = oracle_lib.lib()
    got = np.array([L.pwno_col_ftoint(np.ascontiguousarray(r).ctypes.data) for r in k["col_in"]], np.uint32)
    assert (got == k["col_out"]).all()
    # spot semantics (util.h:48-59): RNE, saturation, NaN/inf -> 0
    f = lambda *v: L.pwno_col_ftoint(np.array(v, np.float32).ctypes.data)
    assert f(0.5 / 255, 1.5 / 255, 2.5 / 255, 0) == 0x00020200
    assert f(30, 30, 0, 0) == 0x0000ffff
    assert f(float("nan"), float("inf"), -1, 1e10) == 0x00000000


def test_normalise_dot_rand(oracle_lib):
    k = np.load(os.path.join(GOLD, "helpers_kat.npz"))
    L = oracle_lib.lib()
    v = k["vec"]
    out = np.zeros(4, np.float32)
    for i in range(len(v)):
        L.pwno_normalise(np.ascontiguousarray(v[i]).ctypes.data, out.ctypes.data)
        assert (out.view(np.uint32) == k["norm"][i].view(np.uint32)).all(), i
        d = L.pwno_dot(np.ascontiguousarray(v[i]).ctypes.data, np.ascontiguousarray(v[(i * 7 + 1) % len(v)]).ctypes.data)
        assert np.float32(d).view(np.uint32) == k["dot"][i].view(np.uint32), i
    for i, s in enumerate(k["seeds"]):
        a = np.array([s], np.uint32)
        r = L.pwno_randfs(a.ctypes.data)
        assert np.float32(r).view(np.uint32) == k["randfs"][i].view(np.uint32)
        assert a[0] == k["seed_after"][i]


def test_upscale(oracle_lib):
    k = np.load(os.path.join(GOLD, "helpers_kat.npz"))
    O = oracle_lib.Oracle()
    up3 = O.upscale(k["up_src"], 3, pitch_bytes=k["up3"].shape[1] * 4)
    assert (up3 == k["up3"]).all()
    assert (O.upscale(k["up_src"], 1) == k["up1"]).all()


def test_blur_seed_skip_ahead(oracle_lib):
    """The blur LCG is affine mod 2^31: 32*g draws = one (A_g, C_g) step.
    This is the identity the GPU blur kernel relies on (post_kernels.hip)."""
    L = oracle_lib.lib()
    A, C = 1, 0
    for g in range(0, 70):
        for cy in (0, 1, 719, 2159, 4319):
            seed0 = (cy * cy + 415135) & 0xFFFFFFFF
            assert L.pwno_blur_seed_at(cy, g) == ((A * seed0 + C) & 0x7FFFFFFF) or g == 0 and L.pwno_blur_seed_at(cy, 0) == seed0
        for _ in range(32):
            A = (A * 25739) & 0x7FFFFFFF
            C = (C * 25739 + 4) & 0x7FFFFFFF


def test_pixel_seed_wraps(oracle_lib):
    """screen.h:19-21 overflows signed int from y = 748 at 4K; uint32 wrap
    (SURVEY.md hard part 6)."""
    L = oracle_lib.lib()
    for (x, y, w) in ((0, 0, 320), (5, 747, 3840), (5, 748, 3840), (3839, 2159, 3840), (7679, 4319, 7680)):
        s = (x + y * y * (w + 1)) & 0xFFFFFFFF
        s = (s * ((s * s) & 0xFFFFFFFF)) & 0xFFFFFFFF
        s = (s * ((s * s) & 0xFFFFFFFF)) & 0xFFFFFFFF
        assert L.pwno_pixel_seed(x, y, w) == s
