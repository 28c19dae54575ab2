"""Device arithmetic primitives against the oracle / KAT fixtures (bit-exact)."""
import os

import numpy as np
import pytest

from conftest import GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import pwnfps_amd
    r = pwnfps_amd.Renderer(64, 64)
    yield r
    r.close()


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def test_rcp_rsqrt_tables(R, oracle_lib):
    from pwnfps_amd import _lib
    L = oracle_lib.lib()
    # every table bucket at several exponents, both signs, plus specials and random bits
    m = (np.arange(2048, dtype=np.uint32) << 12)
    xs = [m | (e << 23) | s for e in (1, 64, 126, 127, 128, 200, 253, 254) for s in (0, 0x80000000)]
    xs.append(np.array([0, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 1, 0x007fffff, 0x00800000, 0x7f7fffff], np.uint32))
    xs.append(np.random.default_rng(1).integers(0, 2 ** 32, 200000, dtype=np.uint32))
    x = np.concatenate(xs).astype(np.uint32)
    for op, fn in ((_lib.PROBE_RCP, L.pwno_rcp), (_lib.PROBE_RSQRT, L.pwno_rsqrt)):
        got = R.probe(op, x)
        want = np.array([fn(float(v)) for v in x[:40000].view(np.float32)], np.float32).view(np.uint32)
        g = got[:40000]
        nan = np.isnan(want.view(np.float32))
        assert (g[~nan] == want[~nan]).all()
        assert np.isnan(g[nan].view(np.float32)).all()


def test_libm_kat_and_oracle(R, oracle_lib):
    from pwnfps_amd import _lib
    k = np.load(os.path.join(GOLD, "libm_kat.npz"))
    assert (R.probe(_lib.PROBE_SINF, k["x_sincos"]) == k["sinf"].view(np.uint32)).all()
    assert (R.probe(_lib.PROBE_COSF, k["x_sincos"]) == k["cosf"].view(np.uint32)).all()
    # the joint form used for the floor normal (one reduction, both polynomials)
    assert (R.probe(_lib.PROBE_SIN_OF_PAIR, k["x_sincos"]) == k["sinf"].view(np.uint32)).all()
    assert (R.probe(_lib.PROBE_COS_OF_PAIR, k["x_sincos"]) == k["cosf"].view(np.uint32)).all()
    # the fixture was produced by glibc in the default MXCSR mode; the reference
    # executable (and the GPU build) flush results below FLT_MIN to zero
    ge = R.probe(_lib.PROBE_EXPF, k["x_exp"])
    norm = np.abs(k["expf"]) >= np.float32(1.17549435e-38)
    assert norm.sum() > 20000 and (~norm).sum() > 100
    assert (ge[norm] == k["expf"].view(np.uint32)[norm]).all()
    assert (ge[~norm] == 0).all()
    # wider sweep against the oracle's restatement (itself pinned to glibc on all floats)
    L = oracle_lib.lib()
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-130, 130, 60000), rng.standard_normal(20000) * 1e4,
                        10.0 ** rng.uniform(-30, 30, 20000), [0.0, -0.0, 0.785398, 0.7853982, 120.0, 119.99999]]).astype(np.float32)
    for op, fn in ((_lib.PROBE_SINF, L.pwno_sinf), (_lib.PROBE_COSF, L.pwno_cosf),
                   (_lib.PROBE_SIN_OF_PAIR, L.pwno_sinf), (_lib.PROBE_COS_OF_PAIR, L.pwno_cosf)):
        want = np.array([fn(float(v)) for v in x], np.float32).view(np.uint32)
        assert (R.probe(op, x) == want).all()
    # the device code tests the argument range once per WAVE (dev_math.h): batches in which every argument is in
    # the middle range 2^-12 <= |y| < 120 take the one-branch path (the batches above almost never do), among
    # them arguments below pi/4, which glibc sends down another path with -- the claim -- the same result
    xm = np.concatenate([rng.uniform(-119.9, 119.9, 40000), rng.uniform(-0.8, 0.8, 20000), 2.0 ** rng.uniform(-12, -1, 3936),
                         [2.0 ** -12, -2.0 ** -12, 0.785398, 0.7853982, -0.7853981, 119.99999, 1.5707964, 3.1415927]]).astype(np.float32)
    xm = xm[(np.abs(xm) >= np.float32(2.0 ** -12)) & (np.abs(xm) < 120)]
    xm = xm[:len(xm) // 64 * 64]
    for op, fn in ((_lib.PROBE_SINF, L.pwno_sinf), (_lib.PROBE_COSF, L.pwno_cosf),
                   (_lib.PROBE_SIN_OF_PAIR, L.pwno_sinf), (_lib.PROBE_COS_OF_PAIR, L.pwno_cosf)):
        want = np.array([fn(float(v)) for v in xm], np.float32).view(np.uint32)
        assert (R.probe(op, xm) == want).all()
    xf = np.concatenate([-rng.uniform(0, 87.0, 30000), rng.uniform(-1, 1, 2000)]).astype(np.float32)     # |x| < 88 throughout, results normal
    want = np.array([L.pwno_expf(float(v)) for v in xf], np.float32)
    assert (R.probe(_lib.PROBE_EXPF, xf) == want.view(np.uint32)).all()
    xe = np.concatenate([-rng.uniform(0, 105, 60000), rng.uniform(-1, 1, 5000), [0.0, -87.0, -88.0, -103.9, -104.0, -1e30]]).astype(np.float32)
    want = np.array([L.pwno_expf(float(v)) for v in xe], np.float32)
    got = R.probe(_lib.PROBE_EXPF, xe).view(np.float32)
    # results below FLT_MIN: the reference executable runs FTZ (crtfastmath), so does the GPU build
    tiny = np.abs(want) < np.float32(1.17549435e-38)
    assert (got[~tiny].view(np.uint32) == want[~tiny].view(np.uint32)).all()
    assert (got[tiny] == 0).all()


def test_sqrt_and_divide_are_ieee(R):
    from pwnfps_amd import _lib
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(0, 4, 100000), 10.0 ** rng.uniform(-30, 30, 100000), [0.0, 1.0, 2.0, 4.0]]).astype(np.float32)
    assert (R.probe(_lib.PROBE_SQRT, x) == np.sqrt(x).view(np.uint32)).all()
    a = (rng.standard_normal(200000) * 10.0 ** rng.uniform(-10, 10, 200000)).astype(np.float32)
    b = (rng.standard_normal(200000) * 10.0 ** rng.uniform(-10, 10, 200000)).astype(np.float32)
    ab = np.stack([a, b], 1).ravel()
    with np.errstate(all="ignore"):
        want = (a / b).astype(np.float32)
    ok = np.abs(want) >= np.float32(1.17549435e-38)
    got = R.probe(_lib.PROBE_DIV, ab)
    assert (got[ok] == want.view(np.uint32)[ok]).all()


def test_ftoint_randfs_kat(R, oracle_lib):
    from pwnfps_amd import _lib
    k = np.load(os.path.join(GOLD, "helpers_kat.npz"))
    assert (R.probe(_lib.PROBE_FTOINT, _f(k["col_in"]).ravel()) == k["col_out"]).all()
    # the pack is v_cvt_pk_u8_f32 + a wave-uniform patch for s >= 2^31 (dev_math.h col_pack4): batches with and
    # without such lanes, ties, NaN, infinities, against the oracle's restatement of util.h:48-59
    L = oracle_lib.lib()
    rng = np.random.default_rng(5)
    tame = np.concatenate([rng.uniform(-0.5, 1.5, (4096, 4)), (rng.integers(-3, 260, (2048, 4)) + 0.5) / 255.0,
                           rng.standard_normal((2048, 4)) * 40]).astype(np.float32)
    wild = tame.copy()
    idx = rng.integers(0, wild.size, 3000)
    wild.reshape(-1)[idx] = rng.choice(np.array([np.nan, np.inf, -np.inf, 8.5e6, 8421505.0, 8421504.0, 1e30, -1e30, 3e38, 2.0 ** 23], np.float32), len(idx))
    for batch in (tame, wild):
        want = np.array([L.pwno_col_ftoint(np.ascontiguousarray(v).ctypes.data) for v in batch], np.uint32)
        assert (R.probe(_lib.PROBE_FTOINT, batch.ravel()) == want).all()
    assert (R.probe(_lib.PROBE_RANDFS, k["seeds"]) == k["randfs"].view(np.uint32)).all()
