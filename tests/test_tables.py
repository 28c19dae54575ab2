"""The rcp/rsqrt approximation tables: fixture == oracle copy == product copy,
and the emulation's known values."""
import os
import re

import numpy as np

from conftest import GOLD, ROOT, host_is_intel


def _parse_c_array(path, name):
    src = open(path).read()
    m = re.search(name + r"\[2048\] = \{(.*?)\};", src, re.S)
    assert m, (path, name)
    return np.array([int(v, 16) for v in re.findall(r"0x[0-9a-fA-F]+", m.group(1))], np.uint16)


def test_three_copies_agree():
    rcp = np.fromfile(os.path.join(GOLD, "rcp_table.u16"), "<u2")
    rsq = np.fromfile(os.path.join(GOLD, "rsqrt_table.u16"), "<u2")
    assert rcp.shape == (2048,) and rsq.shape == (2048,)
    o = os.path.join(ROOT, "oracle", "approx_tables.h")
    p = os.path.join(ROOT, "pwnfps_amd", "csrc", "approx_tables.inc")
    assert (_parse_c_array(o, "pwn_rcp_tab") == rcp).all()
    assert (_parse_c_array(o, "pwn_rsqrt_tab") == rsq).all()
    assert (_parse_c_array(p, "pwn_host_rcp_tab") == rcp).all()
    assert (_parse_c_array(p, "pwn_host_rsqrt_tab") == rsq).all()


def _bits(f):
    return int(np.array([f], np.float32).view(np.uint32)[0])


def test_known_values(oracle_lib):
    L = oracle_lib.lib()
    assert _bits(L.pwno_rcp(1.0)) == 0x3f7ff000
    assert _bits(L.pwno_rcp(3.0)) == 0x3eaaa000
    assert _bits(L.pwno_rsqrt(1.0)) == 0x3f7ff000
    assert _bits(L.pwno_rsqrt(2.0)) == 0x3f34f800
    # exponent invariance and sign handling
    assert _bits(L.pwno_rcp(-6.0)) == (0x3eaaa000 - (1 << 23)) | 0x80000000
    assert _bits(L.pwno_rsqrt(8.0)) == 0x3f34f800 - (1 << 23)
    assert np.isinf(L.pwno_rcp(0.0)) and np.isinf(L.pwno_rsqrt(0.0))
    assert L.pwno_rcp(float("inf")) == 0.0 and L.pwno_rsqrt(float("inf")) == 0.0
    assert np.isnan(L.pwno_rsqrt(-1.0))


def test_tables_match_this_cpu_if_intel(oracle_lib):
    """On an Intel host the emulation must equal the hardware instructions
    (the reference harness's native build exposes them)."""
    import refharness
    if not (host_is_intel() and refharness.available("hw")):
        import pytest
        pytest.skip("needs an Intel host and oracle/_ref/libpwnref_hw.so")
    R = refharness.RefHarness("hw")
    L = oracle_lib.lib()
    rng = np.random.default_rng(5)
    xs = (rng.standard_normal(20000) * 10.0 ** rng.uniform(-20, 20, 20000)).astype(np.float32)
    for x in xs:
        x = float(x)
        assert _bits(R.lib.pwnref_rcp(x)) == _bits(L.pwno_rcp(x))
        if x > 0:
            assert _bits(R.lib.pwnref_rsqrt(x)) == _bits(L.pwno_rsqrt(x))
