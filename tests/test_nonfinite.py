"""Pixels whose arithmetic leaves the finite range.

A ray that enters a ramp with ray.y == 0.5*ray.x (or ray.z) exactly is tilted to
ray.y == 0 and trace.h:461 divides by zero; depth becomes -inf and NaNs reach the
colour.  Generic cameras never do this, axis-aligned ones on lattice positions do
(tools/fuzz_parity.py --lattice).  The reference is built with -ffast-math, i.e.
-ffinite-math-only, so at exactly these pixels its output depends on what the
compiler assumed; the same sources built with -fno-finite-math-only show the plain
SSE/IEEE outcome, and that is what the oracle and the HIP path reproduce -- bit for
bit, alpha byte included.  tests/golden/nonfinite/*.npz hold three such scenes with
both reference renderings (tools/fuzz_parity.py ... --keep-nonfinite).
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLD

SCENES = sorted(glob.glob(os.path.join(GOLD, "nonfinite", "*.npz")))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_scenes_present():
    assert len(SCENES) == 3


@pytest.mark.parametrize("path", SCENES, ids=[os.path.basename(p) for p in SCENES])
def test_oracle_follows_the_ieee_build(path, oracle_lib):
    d = np.load(path)
    O = oracle_lib.Oracle()
    O.load_level_text(str(d["text"]))
    O.set_spheres(d["sph"])
    sb, z = O.render(int(d["w"]), int(d["h"]), d["cam"], sec=float(d["sec"]), blur=int(d["blur"]))
    assert (~np.isfinite(z)).sum() > 0
    assert (sb == d["ref_nf"]).all() and (bits(z) == bits(d["ref_nf_z"])).all()
    assert (sb != d["ref_shipped"]).any()      # the shipped-flags build really differs here


@pytest.mark.gpu
@pytest.mark.parametrize("path", SCENES, ids=[os.path.basename(p) for p in SCENES])
def test_hip_path_follows_the_ieee_build(path):
    import pwnfps_amd
    d = np.load(path)
    r = pwnfps_amd.Renderer(int(d["w"]), int(d["h"]))
    r.level_load_text(str(d["text"]))
    r.set_objects(d["sph"])
    r.set_blur_passes(int(d["blur"]))
    sb, z = r.trace_screen_centred(d["cam"], float(d["sec"]))
    assert (sb == d["ref_nf"]).all(), np.argwhere(sb != d["ref_nf"])[:4].tolist()
    assert (bits(z) == bits(d["ref_nf_z"])).all()
    r.close()
