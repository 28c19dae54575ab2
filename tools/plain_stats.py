import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import pwnfps_amd
gold = "/root/repo/tests/golden"
for level, w, h in [("pwnfps_level", 3840, 2160), ("synth64", 1920, 1080), ("synth256", 7680, 4320)]:
    r = pwnfps_amd.Renderer(w, h)
    r.level_load(os.path.join(gold, "levels", level + ".txt"))
    sph = np.load(os.path.join(gold, "spheres_t0.npy" if level == "pwnfps_level" else os.path.join("levels", level + "_spheres.npy")))
    r.set_objects(sph); r.set_blur_passes(0)
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn)
    if level != "pwnfps_level":
        cam = np.load(os.path.join(gold, "levels", level + "_cams.npy"))[0]
    r.set_counters(True)
    r.trace_screen_centred(cam, 0.0, want_z=False)
    st = r.stats()
    print(level, "wave iterations", st["wave_steps"], "no lane in (sphere|non-room):", round(st["phase_passes"] / st["wave_steps"], 3),
          " and none in 2-high/dq either:", round(st["phase_lanes"] / st["wave_steps"], 3), "paths", [round(x / st["wave_steps"], 3) for x in st["wave_paths"]])
