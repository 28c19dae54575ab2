"""libpwnhip.so must load on a CPU-only box, export every symbol that
include/pwnhip.h declares, and fail loudly (never fall back) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLD, ROOT, level_path, load_spheres

LIB = os.path.join(ROOT, "pwnfps_amd", "libpwnhip.so")


def _build():
    if not os.path.exists(LIB):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pwnfps_amd", "csrc")])


def test_exports_match_header():
    _build()
    hdr = open(os.path.join(ROOT, "include", "pwnhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pwn_[a-z_0-9]+)\s*\(", hdr))
    assert {"pwn_init", "pwn_trace_screen_centred", "pwn_trace_rows_device", "pwn_blur_rows_device",
            "pwn_upload_spheres", "pwn_level_load", "pwn_screen_upscale", "pwn_get_stats"} <= declared
    lib = C.CDLL(LIB)
    for name in sorted(declared):
        assert hasattr(lib, name), "libpwnhip.so does not export " + name
    # the python binding covers the same set
    from pwnfps_amd import _lib
    assert {n for n, _, _ in _lib.ABI} == declared


def test_no_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pwnfps_amd
    with pytest.raises(pwnfps_amd.PwnError) as e:
        pwnfps_amd.Renderer(320, 240)
    assert e.value.code == -2          # PWN_ENODEV
    from pwnfps_amd.dist import HipStripBackend
    with pytest.raises(RuntimeError):
        HipStripBackend(None)


def test_product_does_not_touch_the_oracle():
    """Nothing under pwnfps_amd/, host/ or include/ may import, include, link or
    dlopen anything under oracle/ (or tests/, or the test-only interpreter tools/minilua.py)."""
    bad = []
    for base in ("pwnfps_amd", "include", "host"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            if "build" in dp or "__pycache__" in dp:
                continue
            for f in fs:
                if not f.endswith((".py", ".c", ".cpp", ".h", ".hip", ".inc", "Makefile")):
                    continue
                src = open(os.path.join(dp, f), errors="replace").read()
                for pat in (r'#include\s*"[^"]*oracle', r"\boracle[/.]", r"libpwnoracle", r"pwno_", r"import oracle",
                            r"refharness", r"libpwnref", r"/root/reference", r"import minilua", r"minilua\."):
                    for m in re.finditer(pat, src):
                        line = src[:m.start()].count("\n") + 1
                        bad.append("%s:%d %s" % (os.path.join(dp, f), line, m.group(0)))
    assert not bad, bad


def test_strerror():
    _build()
    lib = C.CDLL(LIB)
    lib.pwn_strerror.restype = C.c_char_p
    assert lib.pwn_strerror(0) == b"ok"
    assert b"HIP device" in lib.pwn_strerror(-2)
    assert b"unknown" in lib.pwn_strerror(-99)


# ---- host-side logic of the product that needs no GPU: the level parser and
# ---- the sphere binning (pwnfps_amd/csrc/level_host.c), against the goldens

class _Portal(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("x1", "z1", "x2", "z2", "rot12", "c1", "c2")]


def _parse(text):
    _build()
    lib = C.CDLL(LIB)
    cells = np.zeros(4096, np.uint8)
    pmap = np.zeros((26, 7), np.int32)
    spawn = np.zeros(2, np.int32)
    if isinstance(text, str):
        text = text.encode("latin-1")
    lib.pwn_parse_level.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.pwn_parse_level(text, len(text), cells.ctypes.data, pmap.ctypes.data, spawn.ctypes.data) == 0
    return cells.reshape(64, 64), pmap, spawn


@pytest.mark.parametrize("name", ["pwnfps_level", "synth64", "synth256"])
def test_product_loader_vs_reference_tables(name):
    t = np.load(os.path.join(GOLD, "levels", name + "_tables.npz"))
    d, p, s = _parse(open(level_path(name), "rb").read())
    assert (d == t["data"]).all() and (p == t["pmap"]).all() and (s == t["spawn"]).all()


def test_product_loader_edge_cases(oracle_lib):
    texts = ["", "..;\r\n\r\n\n.*\r\n;;;;", ";" * 64 + "\n" + "#" * 3 + "\n", "$\n" * 70,
             ".....\n.;a;.\n.;A;.\n..z..\n", ".;C;C;C;.\n", "\n\n\r\r", ";" * 5000,
             "." * 63 + "A\n" + ";" + "." * 62 + "A\n",          # portals in the last column
             "y;Y\n", "ab;\n;BA\n"]
    for t in texts:
        O = oracle_lib.Oracle()
        O.load_level_text(t)
        d0, p0, s0 = O.get_level()
        d, p, s = _parse(t)
        assert (d == d0).all() and (p == p0).all() and (s == s0).all(), repr(t[:20])


def test_product_binning_vs_reference_bins():
    _build()
    lib = C.CDLL(LIB)
    lib.pwn_bin_spheres.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    for name, key in (("pwnfps_level", "t0"), ("synth64", "synth64"), ("synth256", "synth256")):
        t = np.load(os.path.join(GOLD, "levels", name + "_tables.npz"))
        sph = np.ascontiguousarray(load_spheres(key))
        off = np.zeros(4097, np.int32)
        n = lib.pwn_bin_spheres(sph.ctypes.data, len(sph), off.ctypes.data, None, 0)
        assert n == int(t["bin_counts"].sum())
        idx = np.zeros(max(n, 1), np.int32)
        assert lib.pwn_bin_spheres(sph.ctypes.data, len(sph), off.ctypes.data, idx.ctypes.data, n) == n
        assert (np.diff(off) == t["bin_counts"]).all()
        assert (idx[:n] == t["bin_idx"]).all()
    # too small a buffer is reported, not overrun
    assert lib.pwn_bin_spheres(sph.ctypes.data, len(sph), off.ctypes.data, idx.ctypes.data, 1) == -1
