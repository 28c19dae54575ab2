"""pwnfps_amd -- MI355X-native implementation of pwnfps's portal ray-march
render path (trace.h / screen.h of fanzyflani/pwnfps) behind a C ABI.

The compute lives in libpwnhip.so (hand-written HIP for gfx950, see csrc/);
this package is the thin host mirror used by bench.py and the tests."""
from .render import (Renderer, PwnError, SPHERE_DTYPE, PORTAL_DTYPE,  # noqa: F401
                     mat4_iden, mat4_roty, mat4_rotx, spawn_camera)
