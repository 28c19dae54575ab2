#!/usr/bin/env python3
"""Re-render scenes saved by tools/fuzz_parity.py (gpurun_out/fuzz_SEED_I.npz) on the
GPU with counters off and on, and compare with the oracle frame stored in the file."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwnfps_amd  # noqa: E402

tot = 0
for f in sorted(glob.glob(os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fuzz_*.npz"))):
    d = np.load(f)
    w, h = int(d["w"]), int(d["h"])
    for cnt in ((True,) if os.environ.get("PWN_REPRO_COUNTERS") else (False,)):
        for wcam in (True,):
            cam = d["cam"].copy()
            if not wcam:
                cam[:, 3] = (0, 0, 0, 1)
            r = pwnfps_amd.Renderer(w, h)
            r.level_load_text(str(d["text"]))
            r.set_objects(d["sph"])
            r.set_blur_passes(0)
            r.set_counters(cnt)
            a, z = r.trace_screen_centred(cam, float(d["sec"]))
            diff = np.argwhere(a != d["ora"])
            print(os.path.basename(os.environ.get("PWNHIP_LIB", "current")), "%s counters=%d wcam=%d: %d pixels differ from the stored oracle frame%s" % (
                os.path.basename(f), cnt, wcam, len(diff), "" if wcam else " (different camera: expected)"), diff[:3].tolist())
            tot += len(diff)
            r.close()

print(os.path.basename(os.environ.get("PWNHIP_LIB", "current")), "TOTAL differing pixels:", tot)
