#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the portal ray-march path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one frame of the workload BASELINE.json quotes the metric on: the
reference's level.txt scene (its 14 game.lua spheres, camera at the spawn
pose, sec_current = 0) at 3840x2160 with the post-process blur on, i.e. one
level_prepare_render() + trace_screen_centred() (main.c:95,107; screen.h:31-124).  With N > 1 the frame is row-tiled
behind the C ABI (pwn_tiled_*, pwnfps_amd/csrc/pwn_tiled.cpp): rank r traces rows
[r*per, (r+1)*per), two grouped RCCL send/recv launches per frame carry the frame's pre-blur halo rows
between neighbour strips and, behind the rank's blur of its strip, the finished strips to rank 0 -- all of it in order on the
frame's own compute stream, frames alternating between two; three frames are in flight.  Level/sphere
tables and all frame buffers are resident in HBM before the timed region; in the
timed region of `value` the frame stays on the device (rank 0's for N > 1).  The rate with every frame handed over to the host (what
trace_screen_centred does with sbuf, main.c:107) is measured in the same run
through the frames-in-flight API and reported as `d2h_inclusive`.

Timing: the K timed steps are one block, bracketed by barrier + synchronize.
A block of 20 frames lasts 9 ms, so blocks are repeated until a second has
been timed and `value` is the MEDIAN block (every block is exactly K steps);
the spread over blocks and over single launches is reported next to it.

Prints ONE JSON line on rank 0.  After the timed region the last frame is
hashed and compared with the golden hash of the compiled reference.

N > 1: the line carries what it takes to read a first multi-GPU run from the driver's output alone --
`tiling.per_rank` (every rank's trace / blur kernel times, the durations of the two grouped exchanges,
host enqueue time per frame, rows and cost of its strip), the moving cuts, and `tiling.sweep`: short
legs in the same run with the trace grid's room for RCCL at 0 / 16 / 64 workgroups, the gather spread over the
ranks (`rotating_root`), equal strips, one compute stream, whole strips instead of the bounded halo, the other
choreography (`choreo_split`: the exchanges on a third stream), and three compute streams.  `transport` says at top level what carried the data; over
the shared-memory fallback the metric string says that the figure is NOT RCCL over xGMI.

N > 1 cannot fail silently (pwnfps_amd/watch.py): every rank marks the stage it is in in the control plane's key-value
store (not a collective: a rank that hangs does not keep the others from reading it), and the bring-up, the headline
leg and the legs after it each run under a deadline (--bringup-timeout, --headline-timeout, --post-timeout).  When one
passes -- or the launcher sends SIGTERM because a rank died -- rank 0 prints the line anyway: "value": null (or the
headline if that was measured), "incomplete": true, "error", and "stage_reached": per rank the stage, how long ago, the
last error.  Every rank then leaves with exit status 3.  The library's own deadlines (pwn_tiled_set_timeouts) are set
below the bench's, so that a communicator that does not come up comes back as an error the ranks can agree on
(fallback to the shared-memory transport, said in the line) before the watchdog has to end the run.  `tiling.preflight`
records, per rank, the devices it sees and can reach directly, the librccl that dlopen resolved and its version.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
TRACE_BYTES_PER_PIXEL = 8      # colour 4 B + depth 4 B written; everything read is in LDS
FRAME_BYTES_PER_PIXEL = 20     # + blur: read colour 4 + read depth 4 + write final 4


def pmc_traffic(kernel, w, h):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary
    (tools/prof_pmc.sh + tools/pmc_summary.py -> profiles/pmc_latest.csv):
    WRITE_SIZE + 2 x FETCH_SIZE, both in KiB, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950.  None if the summary is absent or
    was taken at another frame size."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.csv")
    try:
        rows = [l.rstrip("\n").rsplit(",", 3) for l in open(path)]
        meta = [l for l in open(path) if l.startswith("#")]
        if not any("%dx%d" % (w, h) in m for m in meta):
            return None
        v = {r[1]: float(r[3]) for r in rows if len(r) == 4 and kernel in r[0] and "<true" not in r[0]}
        return int((v["WRITE_SIZE"] + 2.0 * v["FETCH_SIZE"]) * 1024.0)
    except (OSError, KeyError, ValueError):
        return None


def pmc_issue_rate(kernel, w, h, launch_ms):
    """What actually bounds the trace kernel: wave-instructions issued per
    SIMD and nanosecond -- instruction counts from the committed PMC summary (they do not
    depend on the run), time from this run's HIP events.  Reference points: a CDNA4 SIMD is
    32 lanes wide, so a wave64 VALU instruction takes 2 cycles = 1.2 per ns at 2.4 GHz
    (MI355X_MICROARCH.md); streams of independent plain VALU instructions were measured at
    0.9-1.0 per ns (tools/ubench/valu_rate.hip, VALU_RATE_CALIBRATE=1: the clock sags under
    pure VALU load), v_cmp / VOP3 v_cndmask at 0.55, SALU alone at 0.56."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.csv")
    try:
        meta = [l for l in open(path) if l.startswith("#")]
        if not any("%dx%d" % (w, h) in m for m in meta):
            return None
        rows = [l.rstrip("\n").rsplit(",", 3) for l in open(path)]
        v = {r[1]: float(r[3]) for r in rows if len(r) == 4 and kernel in r[0] and "<true" not in r[0]}
        scalar = v["SQ_INSTS_SALU"] + v["SQ_INSTS_BRANCH"]
        insts = v["SQ_INSTS_VALU"] + scalar + v["SQ_INSTS_LDS"]
        per = 1.0 / (1024.0 * launch_ms * 1e6)          # per SIMD and ns
        out = {"wave_instructions_per_launch": int(insts), "simds": 1024,
               "valu_per_ns_per_simd": round(v["SQ_INSTS_VALU"] * per, 3),
               "valu_architectural_peak_per_ns_per_simd": 1.2, "valu_microbenchmark_per_ns_per_simd": 1.0,
               "scalar_per_ns_per_simd": round(scalar * per, 3),
               "all_per_ns_per_simd": round(insts * per, 3),
               "active_lanes_per_valu_instruction": round(v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"], 1)
               if "SQ_ACTIVE_INST_VALU" in v else None}
        if "SQ_WAVE_CYCLES" in v and "SQ_WAVES" in v and "GRBM_GUI_ACTIVE" in v:
            # share of the kernel's duration the average wave is resident (quad-cycles; GRBM sums 8 XCDs)
            out["mean_wave_residency"] = round(4.0 * v["SQ_WAVE_CYCLES"] / (v["SQ_WAVES"] * v["GRBM_GUI_ACTIVE"] / 8.0), 3)
        return out
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def valu_fractions(kernel, w, h, launch_ms):
    """The trace kernel against the ARCHITECTURAL VALU rate (MI355X_MICROARCH.md: a SIMD is 32 lanes wide, a wave64 VALU
    instruction takes two cycles: 1.2 per ns and SIMD at 2.4 GHz; 1024 SIMDs):
      valu_issue_frac_of_peak = SQ_INSTS_VALU / (1024 x launch ns x 1.2)      how busy the VALU issue ports are
      lane_slot_frac          = that x mean active lanes per VALU instruction / 64   ... with useful lanes
    Instruction counts from the committed PMC summary (profiles/pmc_latest.csv, they do not depend on the run), the
    launch time from this run's HIP events.  None when the summary was taken at another frame size."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.csv")
    try:
        meta = [l for l in open(path) if l.startswith("#")]
        if not any("%dx%d" % (w, h) in m for m in meta) or launch_ms <= 0:
            return None
        rows = [l.rstrip("\n").rsplit(",", 3) for l in open(path)]
        v = {r[1]: float(r[3]) for r in rows if len(r) == 4 and kernel in r[0] and "<true" not in r[0]}
        frac = v["SQ_INSTS_VALU"] / (1024.0 * launch_ms * 1e6 * 1.2)
        lanes = v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"]
        return {"valu_issue_frac_of_peak": round(frac, 4), "lane_slot_frac": round(frac * lanes / 64.0, 4),
                "active_lanes_per_valu_instruction": round(lanes, 2), "sq_insts_valu_per_launch": int(v["SQ_INSTS_VALU"]),
                "peak_valu_per_ns_per_simd": 1.2, "simds": 1024, "counters_from": "profiles/pmc_latest.csv"}
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def nonfinite_scenes(pwnfps_amd):
    """where the reference's arithmetic leaves the finite range (a ramp whose tilt cancels ray.y, trace.h:461) the
    contract is the IEEE build everywhere and the shipped-flags build wherever depth is finite.  The three scenes the
    repo holds with BOTH reference renderings (tests/golden/nonfinite, from tools/fuzz_parity.py --keep-nonfinite),
    through the HIP path, outside every timed region."""
    import glob
    out = []
    for path in sorted(glob.glob(os.path.join(GOLD, "nonfinite", "*.npz"))):
        d = np.load(path)
        r = pwnfps_amd.Renderer(int(d["w"]), int(d["h"]))
        try:
            r.level_load_text(str(d["text"]))
            r.set_objects(d["sph"])
            r.set_blur_passes(int(d["blur"]))
            sb, z = r.trace_screen_centred(d["cam"], float(d["sec"]))
        finally:
            r.close()
        fin = np.isfinite(d["ref_nf_z"])
        out.append({"scene": os.path.basename(path), "pixels": int(sb.size),
                    "equals_ieee_build": bool((sb == d["ref_nf"]).all() and (z.view(np.uint32) == d["ref_nf_z"].view(np.uint32)).all()),
                    "pixels_with_nonfinite_depth": int((~fin).sum()),
                    "pixels_differing_from_the_shipped_flags_build": int((sb != d["ref_shipped"]).sum()),
                    "blur": int(d["blur"])})
    return out


def model_over_measured(w, h, level, launch_ms):
    """VALU issue time of one launch by the committed issue model (tools/issue_model.py -> profiles/r5_issue_model.json) over the
    measured launch time.  NOT a roofline fraction: the model's costs per opcode class come from this repo's own
    microbenchmark, and a value near 1 says "this instruction stream has no stall slack", not "no faster kernel exists"
    (the architectural fractions are roofline.valu_issue_frac_of_peak / lane_slot_frac).  None when the model was not made
    for this frame."""
    try:
        with open(os.path.join(ROOT, "profiles", "r5_issue_model.json")) as f:
            m = json.load(f)
        for c in m["cases"]:
            if (c["w"], c["h"], c["level"]) == (w, h, level) and launch_ms > 0:
                return {"valu_issue_ms_model": c["valu_issue_ms"], "ratio": round(c["valu_issue_ms"] / launch_ms, 4),
                        "model": "profiles/r5_issue_model.txt",
                        "what": "the builder's cost model over the measured launch; > 1 only says the cost table over-predicts"}
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(w, h, cam, spheres, level_file, target_s=10.0):
    """The reference's own code (oracle/_ref, built from /root/reference with the
    reference's flags) timed on this host's cores; falls back to the port."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    threads = os.cpu_count() or 1
    try:
        import refharness
        if refharness.available("hw"):
            R = refharness.RefHarness("hw")
            R.load_level(level_file)
            R.set_spheres(spheres)
            R.render(w, h // 8, cam, threads=threads, want_z=False)       # page in, spin up the team
            t0 = time.perf_counter()
            R.render(w, h, cam, threads=threads, want_z=False)
            t1 = time.perf_counter() - t0
            reps = int(min(20, max(1, round(target_s / max(t1, 1e-3)))))
            best = t1
            for _ in range(reps):
                t0 = time.perf_counter()
                R.render(w, h, cam, threads=threads, want_z=False)
                best = min(best, time.perf_counter() - t0)
            # one thread as well: two frames at about 3 Mpixels/s
            t0 = time.perf_counter()
            R.render(w, h, cam, threads=1, want_z=False)
            one = time.perf_counter() - t0
            t0 = time.perf_counter()
            R.render(w, h, cam, threads=1, want_z=False)
            one = min(one, time.perf_counter() - t0)
            return {"value": round(w * h / best / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "kind": "reference",
                    "value_1_thread": round(w * h / one / 1e6, 3),
                    "sample": "%d full %dx%d frames (trace+blur) of the same scene, best of %d, reference "
                              "sources compiled with its own flags (gcc -O3 -fopenmp -ffast-math -funroll-loops), "
                              "OpenMP over 32-row chunks as screen.h:63" % (reps + 1, w, h, reps + 1)}
    except Exception as e:  # noqa: BLE001 -- a broken checker must not kill the bench line
        sys.stderr.write("cpu_baseline: reference build unusable (%s); using the port\n" % e)
    import oracle
    O = oracle.Oracle()
    O.load_level(level_file)
    O.set_spheres(spheres)
    t0 = time.perf_counter()
    O.render(w, h, cam, threads=threads)
    t1 = time.perf_counter() - t0
    reps = int(min(10, max(1, round(target_s / max(t1, 1e-3)))))
    best = t1
    for _ in range(reps):
        t0 = time.perf_counter()
        O.render(w, h, cam, threads=threads)
        best = min(best, time.perf_counter() - t0)
    return {"value": round(w * h / best / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "sample": "%d full %dx%d frames (trace+blur) of the same scene, best of %d, scalar C restatement "
                      "with table-emulated rcpps/rsqrtps, OpenMP over rows" % (reps + 1, w, h, reps + 1)}


def d2h_leg_one_gpu(r, args, w, h, cam, sec, spheres, blocking_best, same_as_resident):
    """The metric as SURVEY.md 8(d) words it, on one GPU: every frame handed over to the host.  Same step as the
    resident loop (re-bin + upload, trace, blur) plus the D2H into the library's pinned sbuf, `slots` frames in
    flight; the same K-step blocks, median block."""
    import torch
    nsl = max(2, min(args.slots, 4))
    r.frames_config(nsl, sbuf=True)
    held = {"f": None}
    early = args.prepare == "early"

    def d2h_block(n):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for f in range(n + nsl - 1):
            if f < n and early:
                r.set_objects(spheres)
            if f >= nsl - 1:
                held["f"] = r.wait_frame((f - nsl + 1) % nsl)
            if f < n:
                if not early:
                    r.set_objects(spheres)
                r.submit_frame(cam, sec, f % nsl)
        return time.perf_counter() - t1
    d2h_block(args.warmup)
    d2h_s = []
    while sum(d2h_s) < args.min_time and len(d2h_s) < 500:
        d2h_s.append(d2h_block(args.steps))
    d2h_dt = float(np.median(d2h_s))
    d2h_ok = bool(same_as_resident(held["f"]["sbuf"])) if same_as_resident is not None else None
    pcie = {"value": round(w * h * args.steps / d2h_dt / 1e6, 3), "unit": "Mpixels/s",
            "ms_per_step": round(d2h_dt / args.steps * 1e3, 4), "frames_in_flight": nsl,
            "blocks": len(d2h_s), "block_ms_p10_p50_p90": [round(float(np.percentile(d2h_s, q)) * 1e3, 3) for q in (10, 50, 90)],
            "bytes_over_pcie_per_frame": 4 * w * h,
            "pcie_gbs": round(4 * w * h * args.steps / d2h_dt / 1e9, 2),
            "blocking_call_mpix_s": round(w * h / blocking_best / 1e6, 2),
            "last_frame_equals_resident_frame": d2h_ok,
            "what": "set_objects + pwn_submit_frame / pwn_wait_frame: trace + blur + D2H of sbuf into pinned host memory; "
                    "blocking_call_mpix_s = one pwn_trace_screen_centred at a time into the host's registered sbuf, in row strips (blocking_call has the other forms)"}
    r.frames_config(0)
    return pcie


def host_sink_leg(r, args, w, h, cam, sec, spheres, rank, world, transport, barrier, max_over_ranks, same_as_resident, mark=lambda *a, **k: None):
    """The same metric with every frame delivered to the HOST on N > 1 GPUs (main.c:107-109 presents
    every frame there): pwn_tiled_host_sink -- every rank copies its finished strip straight into ONE frame in POSIX
    shared memory over its own PCIe link, there is no gather to rank 0.  Collective: every rank calls it."""
    import torch
    import torch.distributed as dist
    import pwnfps_amd
    import mmap
    mark("host_sink: shutdown of the resident tiling")
    barrier()
    r.tiled_shutdown()
    shm_name = [("/dev/shm/pwn_bench_frames_%d_%d" % (os.getpid(), int(time.time() * 1e3))) if rank == 0 else None]
    if rank == 0:
        with open(shm_name[0], "wb") as f:
            f.truncate(_lib_slots() * 4 * w * h)
    uid2 = [pwnfps_amd.Renderer.tiled_unique_id(transport) if rank == 0 else None]
    dist.broadcast_object_list(shm_name, src=0)
    dist.broadcast_object_list(uid2, src=0)
    fd = os.open(shm_name[0], os.O_RDWR)
    frames_mm = mmap.mmap(fd, _lib_slots() * 4 * w * h)
    os.close(fd)
    mark("host_sink: tiled_init", transport=transport)
    r.tiled_init(rank, world, uid2[0], transport, args.halo)
    mark("host_sink: frames", transport=transport)
    sink_err = None
    try:
        r.tiled_host_sink(frames_mm)
    except Exception as e:                                       # noqa: BLE001 -- reported in the line
        sink_err = "rank %d: %s" % (rank, e)
    ok = torch.tensor([0.0 if sink_err else 1.0], dtype=torch.float64)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    sink_ok = float(ok.item()) == 1.0                            # every rank takes the same branch
    held = {"f": None}
    early = args.prepare == "early"

    def host_block(n):
        barrier()
        t1 = time.perf_counter()
        for i in range(n):
            if not early or i == 0:
                r.set_objects(spheres)
            r.tiled_submit(cam, sec)
            if early and i + 1 < n:
                r.set_objects(spheres)
            if i >= 2:
                held["f"] = r.tiled_wait()
        for _ in range(min(n, 2)):
            held["f"] = r.tiled_wait()
        barrier()
        return max_over_ranks(time.perf_counter() - t1)
    if not sink_ok:
        pcie = {"value": None, "error": sink_err or "pwn_tiled_host_sink failed on another rank"}
    else:
        host_block(args.warmup)
        host_s = []
        while True:
            host_s.append(host_block(args.steps))
            if sum(host_s) >= args.min_time or len(host_s) >= 500:
                break
        host_dt = float(np.median(host_s))
        hinfo = r.tiled_info()
        host_ok = bool(same_as_resident(held["f"]["sbuf"])) if same_as_resident is not None else None
        pcie = {"value": round(w * h * args.steps / host_dt / 1e6, 3), "unit": "Mpixels/s",
                "ms_per_step": round(host_dt / args.steps * 1e3, 4), "frames_in_flight": 3,
                "blocks": len(host_s), "block_ms_p10_p50_p90": [round(float(np.percentile(host_s, q)) * 1e3, 3) for q in (10, 50, 90)],
                "bytes_over_pcie_per_frame_and_rank": int(hinfo["bytes_to_host"] // max(hinfo["frames"] + hinfo["frames_redone"], 1)),
                "pcie_links": world,
                "host_frame_gbs": round(4 * w * h * args.steps / host_dt / 1e9, 2),
                "frames_repeated_with_whole_strips": int(max_over_ranks(float(hinfo["frames_redone"]))),
                "last_frame_equals_resident_frame": host_ok,
                "what": "pwn_tiled_host_sink: set_objects + pwn_tiled_submit / pwn_tiled_wait; every rank copies its finished strip "
                        "into one frame in POSIX shared memory over its own PCIe link, no gather to rank 0"}
    barrier()
    held["f"] = None
    r.tiled_shutdown()
    if rank == 0:
        try:
            os.unlink(shm_name[0])
        except OSError:
            pass
    return pcie


def single_process(args):
    """--single-process: ONE process, ONE handle (pwn_init_multi), the frame row-tiled over --gpus devices by the library's own
    member threads -- what a host shaped like the reference's (one loop, main.c:93-109) gets.  The same step, the same K-step
    blocks and median as the one-process-per-GPU run; `value` = frames that stay on the devices (gathered on device 0),
    `d2h_inclusive` = every frame delivered to the host, every device copying its strip over its own PCIe link."""
    import torch
    import pwnfps_amd
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: libpwnhip.so has no CPU fallback")
    n = args.gpus
    ndev = torch.cuda.device_count()
    # test hook: PWN_BENCH_ONE_DEVICE=1 puts every member on device 0 (the in-process transport between them)
    devices = [0] * n if os.environ.get("PWN_BENCH_ONE_DEVICE") else list(range(n))
    if max(devices) >= ndev:
        sys.exit("bench.py --single-process --gpus %d: this process sees %d device(s)" % (n, ndev))
    w, h = args.width, args.height
    level_file = os.path.join(GOLD, "levels", args.level + ".txt")
    spheres = np.load(os.path.join(GOLD, "spheres_t0.npy")) if args.level == "pwnfps_level" else np.load(os.path.join(GOLD, "levels", args.level + "_spheres.npy"))
    r = pwnfps_amd.Renderer(w, h, devices=devices)
    r.level_load(level_file)
    r.set_objects(spheres)
    r.set_blur_passes(args.blur)
    r.set_frame_timing(max(1, args.time_every))
    r.tiled_set_timeouts(max(5.0, min(120.0, args.bringup_timeout * 0.4)), max(3.0, min(30.0, args.headline_timeout * 0.25)))
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn)
    sec = 0.0
    early = args.prepare == "early"
    nsl = 3
    launch_ms = []
    last = {"f": None}

    def run(k):
        for i in range(k):
            s = i % nsl
            if early:
                r.set_objects(spheres)
            if i >= nsl:
                last["f"] = r.wait_frame(s)
                if last["f"]["timed"]:
                    launch_ms.append(last["f"]["trace_ms"])
            if not early:
                r.set_objects(spheres)
            r.submit_frame(cam, sec, s)
        for i in range(max(0, k - nsl), k):
            last["f"] = r.wait_frame(i % nsl)

    def leg(warmup, min_time):
        run(warmup)
        launch_ms.clear()
        blocks = []
        while sum(blocks) < min_time and len(blocks) < 500:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            blocks.append(time.perf_counter() - t0)
        return float(np.median(blocks)), blocks
    r.frames_config(nsl, sbuf=False)
    dt, block_s = leg(args.warmup, args.min_time)
    one = {"members": 1, "transport": None, "devices": devices, "cuts": [0, h], "halo_rows": 0, "host_sink": False, "frames": 0, "frames_redone": 0, "recuts": 0, "note": ""}
    gi = r.group_info() if n > 1 else one
    trace_ms = float(np.mean(launch_ms)) if launch_ms else 0.0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    frame_hash, parity = None, None
    try:
        import oracle
        frame_hash = oracle.fnv64(r.read_plane(last["f"]["d_sbuf"]))
        with open(os.path.join(GOLD, "frames.json")) as f:
            want = [c for c in json.load(f)["cases"] if c["level"] == args.level and (c["w"], c["h"]) == (w, h) and c["sec"] == 0.0
                    and c["nspheres"] == len(spheres) and c["name"].startswith(("level_spawn", args.level + "_cam0"))]
        if want and args.level == "pwnfps_level":
            parity = bool(frame_hash == (want[0]["post"] if args.blur else want[0]["pre"]))
    except Exception as e:                                           # noqa: BLE001
        sys.stderr.write("parity check skipped: %s\n" % e)
    # ---- every frame delivered to the host
    pcie = None
    if not args.no_d2h:
        r.frames_config(nsl, sbuf=True)
        dth, bh = leg(max(2, args.warmup // 2), args.min_time)
        same = bool(oracle.fnv64(last["f"]["sbuf"]) == frame_hash) if frame_hash is not None else None
        sb = np.zeros((h, w), np.uint32)
        r.frames_config(0)
        r.host_register(sb)
        best = 1e9
        for _ in range(7):
            t1 = time.perf_counter()
            r.trace_screen_centred(cam, sec, want_z=False, sbuf=sb)
            best = min(best, time.perf_counter() - t1)
        same_b = bool(oracle.fnv64(sb) == frame_hash) if frame_hash is not None else None
        r.host_unregister(sb)
        pcie = {"value": round(w * h * args.steps / dth / 1e6, 3), "unit": "Mpixels/s", "ms_per_step": round(dth / args.steps * 1e3, 4),
                "frames_in_flight": nsl, "blocks": len(bh), "pcie_links": len(set(devices)), "host_frame_gbs": round(4 * w * h * args.steps / dth / 1e9, 2),
                "last_frame_equals_resident_frame": same,
                "blocking_call_mpix_s": round(w * h / best / 1e6, 2), "blocking_call_ms": round(best * 1e3, 4), "blocking_call_frame_equals_resident_frame": same_b,
                "what": "pwn_frames_config(PWN_FRAME_SBUF) on the group's handle: every member copies its finished strip into the group's pinned frame "
                        "over its own device's PCIe link; blocking_call = one pwn_trace_screen_centred at a time into the host's registered sbuf"}
    gi2 = r.group_info() if n > 1 else one
    pix = w * h
    rows = [gi["cuts"][i + 1] - gi["cuts"][i] for i in range(n)]
    strip_pix = max(rows) * w
    achieved = TRACE_BYTES_PER_PIXEL * strip_pix / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
    one_device = len(set(devices)) == 1 and n > 1
    line = {
        "metric": ("Mpixels/s at %dx%d (level.txt scene, trace + blur), frames resident on the devices (gathered on device 0); ONE process, one handle "
                   "(pwn_init_multi), %d member threads" % (w, h, n))
                  + (" -- ALL MEMBERS ON ONE DEVICE (test hook): not a multi-GPU figure" if one_device else ""),
        "value": round(pix * args.steps / dt / 1e6, 3), "unit": "Mpixels/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "single_process": True, "transport": gi["transport"], "transport_note": gi.get("note") or None, "devices": devices,
        "timing": {"blocks_of_k_steps": len(block_s), "value_is": "median block",
                   "block_ms_p10_p50_p90": [round(float(np.percentile(block_s, q)) * 1e3, 4) for q in (10, 50, 90)]},
        "config": {"workload": "pwnfps level.txt scene (14 game.lua spheres, spawn pose, sec_current=0), %dx%d, POSTPROC_BLUR=%d, one frame per step" % (w, h, args.blur),
                   "level": args.level, "width": w, "height": h, "blur_passes": args.blur,
                   "parallelism": "rows/%d with moving cuts inside ONE process: a library-owned thread per device drives pwn_tiled.cpp's in-stream choreography, "
                                  "exchange over %s, 3 frames in flight" % (n, gi["transport"]),
                   "host_loop": "set_objects(i) / wait for a free slot / submit(i) on the one handle"},
        "tiling": {"cuts": gi["cuts"], "halo_rows": gi["halo_rows"], "frames": gi2["frames"], "frames_redone": gi2["frames_redone"], "recuts": gi2["recuts"]},
        "roofline": {"bound": "hbm", "kernel": "pwn_trace_kernel", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "algorithmic_bytes_per_launch": TRACE_BYTES_PER_PIXEL * strip_pix,
                     "bytes_per_pixel": TRACE_BYTES_PER_PIXEL, "pixels_per_launch": strip_pix, "avg_launch_ms": round(trace_ms, 4),
                     "note": "the slowest member's strip, HIP events on its launch stream (the launches share their chip with the neighbour frames' kernels)"},
        "frame_bytes_per_pixel": FRAME_BYTES_PER_PIXEL, "frame_gbs": round(FRAME_BYTES_PER_PIXEL * pix * args.steps / dt / 1e9, 3),
        "parity_vs_reference_golden": parity, "frame_fnv64": frame_hash,
    }
    if pcie:
        line["d2h_inclusive"] = pcie
    print(json.dumps(line), flush=True)
    r.close()


def _lib_slots():
    from pwnfps_amd import _lib
    return _lib.PWN_TILED_SLOTS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--level", default="pwnfps_level")
    ap.add_argument("--blur", type=int, default=1)
    ap.add_argument("--min-time", type=float, default=3.0, help="repeat the K-step block until this many seconds were timed (each of the "
                    "resident and the d2h leg: the GPU is busy for >= 6 s of a default run)")
    ap.add_argument("--sweep-time", type=float, default=0.3, help="N > 1: seconds timed per leg of tiling.sweep (0 = no sweep)")
    ap.add_argument("--sweep-budget", type=float, default=100.0,
                    help="N > 1: seconds the sweep legs that set the tiling up again may take together; the legs left over are marked skipped")
    ap.add_argument("--sweep-nonblocking", action="store_true",
                    help="N > 1 over RCCL: one more sweep leg with the communicator driven non-blocking (PWN_TILED_RCCL_MODE=nonblocking)")
    ap.add_argument("--bringup-timeout", type=float, default=300.0,
                    help="N > 1: seconds the bring-up may take (control plane, preflight, communicator, four frames through every leg of the "
                         "exchange; a fallback to the shared-memory transport starts the clock again); then rank 0 prints a diagnostic line "
                         "(value null, the stage every rank reached) and every rank leaves with status 3")
    ap.add_argument("--headline-timeout", type=float, default=120.0, help="N > 1: the same for the headline leg (warm-up + the timed blocks)")
    ap.add_argument("--post-timeout", type=float, default=240.0,
                    help="N > 1: seconds the legs AFTER the headline (sweep, host-sink leg) may take in all; then every rank stops and "
                         "rank 0 prints the line with what it has -- a hang in an extra leg must not cost the headline number")
    ap.add_argument("--slots", type=int, default=3, help="frames in flight of the d2h_inclusive leg")
    ap.add_argument("--resident-slots", type=int, default=3, help="N = 1: frames in flight of the resident loop (1 = strictly one after the other)")
    ap.add_argument("--scheduler", choices=["units", "refill"], default=None, help="trace kernel scheduler (default: the library's)")
    ap.add_argument("--time-every", type=int, default=8,
                    help="HIP events around the trace kernel of every N-th frame of the timed region (an event between two "
                         "kernels costs ~4 us of pipeline; 1 = every launch)")
    ap.add_argument("--halo", type=int, default=-1, help="N > 1: pre-blur rows exchanged with each neighbour strip (-1 default, 0 whole strips)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prepare", choices=("early", "late"), default="early",
                    help="set_objects of a frame before (early) or after (late) the host waits for a free slot")
    ap.add_argument("--trace-room", type=int, default=None, help="PWN_OPT_TRACE_ROOM (default: the library's, -1 = it measures)")
    ap.add_argument("--one-stream", action="store_true",
                    help="N = 1: PWN_OPT_FRAME_OVERLAP 0 for the whole run -- every launch by itself, for rocprofv3 --kernel-trace runs whose "
                         "average kernel durations are to be compared with roofline.avg_launch_ms")
    ap.add_argument("--no-d2h", action="store_true",
                    help="skip the d2h_inclusive leg: for rocprofv3 --kernel-trace runs, where the profiler serialises the copies of "
                         "that leg with the kernels and their durations (2.7x) would be averaged into the resident loop's")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N in ONE process: one handle (pwn_init_multi), the library's member threads row-tile every frame over devices 0..N-1; "
                         "started plainly (python bench.py --gpus N --single-process), not under torch.distributed.run")
    args = ap.parse_args()
    if args.single_process:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            sys.exit("bench.py --single-process is one process: start it without torch.distributed.run")
        return single_process(args)

    # (multi-process GPU work on this pool: the host driver only supports dmabuf IPC; already exported where the driver runs this)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import pwnfps_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: libpwnhip.so has no CPU fallback")
    # test hooks (a 1-GPU box cannot run RCCL with 2 ranks): PWN_BENCH_ONE_DEVICE=1 puts every rank on
    # device 0 and PWN_BENCH_TRANSPORT=shm moves the messages through shared memory; the driver sets neither
    if os.environ.get("PWN_BENCH_ONE_DEVICE"):
        local = 0
    transport = os.environ.get("PWN_BENCH_TRANSPORT", "rccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    line_out = sys.stdout
    if world > 1:
        # stdout carries the JSON line and nothing else: from here on whatever libraries write to descriptor 1 (gloo announces its
        # connections there) goes to stderr, and the line is written to a copy of the original descriptor
        sys.stdout.flush()
        line_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        # control plane (the group id, barriers, the max over ranks): gloo.  The data path is the
        # library's own RCCL communicator (ncclSend / ncclRecv over xGMI), not torch's.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    from pwnfps_amd import watch
    board = watch.Board(rank, world, watch.default_store() if world > 1 else None)
    partial = {"line": None}             # what a diagnostic line can already say (filled as the run proceeds)

    def diagnostic_line(reason, stages):
        """rank 0, a deadline passed or the launcher sent SIGTERM: the line with what there is (no GPU call, no collective)"""
        if partial.get("build") is not None:
            try:
                line = partial["build"]()            # the headline was measured: the whole line, with what the later legs have so far
            except Exception as e:                   # noqa: BLE001 -- the diagnostic must come out whatever state the run is in
                line = dict(partial["line"] or {}, line_error=str(e))
        else:
            line = dict(partial["line"] or {})
        line.setdefault("metric", "Mpixels/s at %dx%d (level.txt scene, trace + blur), frames resident on the device" % (args.width, args.height))
        line.setdefault("value", None)
        line.setdefault("unit", "Mpixels/s")
        line.update(n_gpus=world, steps=args.steps, warmup=args.warmup, higher_is_better=True, incomplete=True, error=reason,
                    stage_reached=stages)
        return line
    dog = watch.Watch(board, diagnostic_line, out=line_out)
    _RUN.update(dog=dog, board=board, world=world, rank=rank)
    if world > 1:
        dog.catch_sigterm()
        board.mark("control_plane_up")
        dog.arm(args.bringup_timeout, "bring-up")

    def die_here(stage):
        """test hook (tests/test_gpu_bench_ranks.py): PWN_BENCH_DIE_AT=STAGE:RANK -- that rank leaves the process at that stage"""
        want = os.environ.get("PWN_BENCH_DIE_AT", "")
        if want and want.split(":")[0] == stage and int(want.split(":")[1]) == rank:
            sys.stdout.flush()
            os._exit(17)

    w, h = args.width, args.height
    level_file = os.path.join(GOLD, "levels", args.level + ".txt")
    if args.level == "pwnfps_level":
        spheres = np.load(os.path.join(GOLD, "spheres_t0.npy"))
    else:
        spheres = np.load(os.path.join(GOLD, "levels", args.level + "_spheres.npy"))

    r = pwnfps_amd.Renderer(w, h, device=local)
    if args.trace_room is not None:
        r.set_trace_room(args.trace_room)
    r.level_load(level_file)
    r.set_objects(spheres)
    r.set_blur_passes(args.blur)
    if args.scheduler:
        r.set_scheduler(args.scheduler)
    r.set_frame_timing(max(1, args.time_every))       # (N = 1 with two compute streams: off in the headline leg, see the roofline leg)
    _, _, spawn = r.get_level()
    cam = pwnfps_amd.spawn_camera(spawn)            # main.c:61-64
    sec = 0.0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # The CPU baseline runs FIRST (rank 0, N = 1): ~15 s of host work, then the GPU legs in one piece (>= 6 s busy).
    cpu_line = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu_line = cpu_baseline(w, h, cam, spheres, level_file)

    nres = max(1, min(args.resident_slots, 4))
    # frames alternate between two compute streams (the library's default, PWN_OPT_FRAME_OVERLAP; PWN_FRAME_OVERLAP=0 in the
    # environment switches it off): the next frame's trace grid fills what the current one's tail leaves idle
    overlap_on = os.environ.get("PWN_FRAME_OVERLAP", "1") not in ("0", "") and nres >= 2 and not args.one_stream
    if args.one_stream:
        r.set_frame_overlap(False)
    tinfo = None
    if world == 1:
        r.frames_config(nres, sbuf=False)
    else:
        # ---- preflight: what every rank sees (devices, direct peer access from its own, the librccl dlopen resolved, its
        # version, how the communicator will be driven), marked on the board -- readable whatever happens next -- and
        # gathered for the line
        # (a first ncclCommInitRank of eight ranks on a cold node can take tens of seconds: a deadline that is too tight would send a
        # healthy run to the shared-memory fallback; 120 s of 300 by default)
        lib_init_s = max(5.0, min(120.0, args.bringup_timeout * 0.4))
        lib_wait_s = max(3.0, min(30.0, args.headline_timeout * 0.25))
        r.tiled_set_timeouts(lib_init_s, lib_wait_s)       # below the bench's own deadlines: an error the ranks can agree on comes first
        try:
            pre_mine = r.tiled_preflight()
        except Exception as e:                                       # noqa: BLE001 -- reported in the line
            pre_mine = {"error": str(e)}
        pre_mine["rank"] = rank
        board.mark("preflight", preflight=pre_mine)
        die_here("preflight")
        preflight = [None] * world
        dist.all_gather_object(preflight, pre_mine)
        partial["line"] = {"tiling": {"preflight": preflight}}
        # (what to read first when a run falls back to the shared-memory transport: a rank whose device cannot reach another rank's
        # directly, ranks that resolved different librccl files)
        try:
            unreachable = [[q["rank"], d] for q in preflight for d, ok in enumerate(q.get("can_access_peer", [])) if ok != 1]
            libs = sorted({str(q.get("librccl")) for q in preflight})
            preflight_summary = {"every_device_reaches_every_other": not unreachable, "unreachable_rank_device_pairs": unreachable[:32],
                                 "librccl_files": libs, "rccl_versions": sorted({q.get("rccl_version") for q in preflight})}
        except Exception as e:                                       # noqa: BLE001
            preflight_summary = {"error": str(e)}
        partial["line"]["tiling"]["preflight_summary"] = preflight_summary
        # Every rank first checks that it can load the transport at all (a rank without librccl would leave the
        # others waiting inside ncclCommInitRank).  If one cannot, all ranks agree on the shared-memory test
        # transport instead -- the same kernels and messages, through the host -- and the line says so.
        transport_note = None
        board.mark("transport_check", transport=transport)
        if transport == "rccl":
            try:
                pwnfps_amd.Renderer.tiled_unique_id("rccl")
                mine = 1.0
            except Exception as e:                                   # noqa: BLE001 -- reported below
                mine, transport_note = 0.0, "librccl could not be loaded on rank %d: %s" % (rank, e)
            ok = torch.tensor([mine], dtype=torch.float64)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0:
                transport = "shm"
                transport_note = transport_note or "librccl could not be loaded on another rank"
                print("bench.py rank %d: %s -- falling back to the shared-memory transport" % (rank, transport_note), file=sys.stderr)

        def bring_up(tp, halo=None, tag=""):
            """Communicator plus four frames through every leg of the exchange (halo group, gather group, the ranks'
            words); every rank learns whether ALL ranks got through.  Each step is marked on the board; an error of
            this rank is marked too, BEFORE the collective that tells the others (which hangs if a peer is gone: the
            watchdog then prints the marks)."""
            board.mark(tag + "unique_id", transport=tp)
            u = [pwnfps_amd.Renderer.tiled_unique_id(tp) if rank == 0 else None]
            dist.broadcast_object_list(u, src=0)
            err = None
            timed_out = 0.0
            try:
                board.mark(tag + "tiled_init", transport=tp, limit_s=lib_init_s)
                die_here(tag + "tiled_init")
                r.tiled_init(rank, world, u[0], tp, args.halo if halo is None else halo)
                board.mark(tag + "first_frames", transport=tp, limit_s=lib_wait_s)
                die_here(tag + "first_frames")
                for i in range(4):
                    r.set_objects(spheres)
                    r.tiled_submit(cam, sec)
                    if i >= 2:
                        r.tiled_wait()
                r.tiled_wait()
                r.tiled_wait()
                torch.cuda.synchronize()
            except Exception as e:                                   # noqa: BLE001 -- reported in the line
                err = "rank %d: %s" % (rank, e)
                timed_out = 1.0 if getattr(e, "code", 0) == pwnfps_amd._lib.PWN_ETIMEDOUT else 0.0
                board.note_error(err)
            board.mark(tag + "agree", transport=tp, error=err)
            good = torch.tensor([0.0 if err else 1.0, -timed_out], dtype=torch.float64)
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
            bring_up.timed_out = float(good[1].item()) < 0.0          # a deadline of the library passed on some rank
            return float(good[0].item()) == 1.0, err
        bring_up.timed_out = False

        up, err = bring_up(transport)
        if not up and transport == "rccl" and bring_up.timed_out:
            # A bring-up that ran into its deadline inside ncclCommInitRank leaves a helper thread behind in RCCL (pwn_tiled.cpp:
            # there is no communicator to abort yet).  Should the missing peer turn up later, that thread finishes the initialisation
            # beside whatever this process does next: no fallback in the same process -- the diagnostic line, and a fresh start.
            board.mark("bring_up_timed_out", error=err or "on another rank", transport=transport)
            dog.bail("the RCCL bring-up ran into the library's deadline (%s): no fallback to the shared-memory transport in a process that may still "
                     "hold a thread inside RCCL -- start again" % (err or "on another rank (see stage_reached)"))
        if not up and transport == "rccl":
            # (an error every rank can return from -- a communicator that cannot be made in this environment, or one that
            # did not come up within the library's deadline.  A peer that DIED leaves the agreement above hanging: that ends
            # in the watchdog's diagnostic line)
            transport_note = "the RCCL transport did not come up (%s)" % (err or "on another rank")
            print("bench.py rank %d: %s -- falling back to the shared-memory transport" % (rank, transport_note), file=sys.stderr)
            r.tiled_shutdown()
            transport = "shm"
            dog.arm(args.bringup_timeout, "bring-up over the shared-memory fallback")
            up, err = bring_up(transport, tag="fallback:")
        if not up:
            board.mark("bring_up_failed", error=err or "error on another rank", transport=transport)
            dog.bail("the row tiling did not come up over %s: %s" % (transport, err or "error on another rank (see stage_reached)"))
        tinfo = r.tiled_info()
        partial["line"]["tiling"].update(transport=transport, transport_note=transport_note,
                                         rccl_nonblocking=tinfo["rccl_nonblocking"], deadlines_s={"library_init": lib_init_s, "library_wait": lib_wait_s,
                                                                                               "bring_up": args.bringup_timeout, "headline": args.headline_timeout, "post": args.post_timeout})
        board.mark("bring_up_done", transport=transport)
        dog.disarm()

    launch_ms = []
    last = {"f": None}
    tiled_depth = [3]
    diag = {"trace_ms": [], "blur_ms": [], "halo_ms": [], "gather_ms": [], "frame_ms": [], "enqueue_us": [], "rows": 0, "cost": 0, "redone": 0}

    def diag_reset():
        for k in diag:
            diag[k] = [] if isinstance(diag[k], list) else 0
        launch_ms.clear()

    def per_rank():
        """every rank's means of what it measured since diag_reset(), gathered over the control plane"""
        mine = {k: (round(float(np.mean(v)), 4) if v else None) for k, v in diag.items() if isinstance(v, list)}
        mine.update(rows=diag["rows"], cost=diag["cost"], frames_redone=diag["redone"], timed_frames=len(diag["trace_ms"]))
        mine.update(trace_room=r.trace_room_state()["room_now"])          # PWN_OPT_TRACE_ROOM as this rank's own measurement left it
        every = [None] * world
        dist.all_gather_object(every, mine)
        return {k: [e[k] for e in every] for k in mine}
    early = args.prepare == "early"

    # Every step re-bins and re-uploads the spheres first, like the reference's frame loop does
    # (level_prepare_render, main.c:95), although this benchmark's spheres do not move.
    def run(n):
        """n frames back to back; nothing is handed to the host"""
        def note(f):
            if f["timed"]:
                launch_ms.append(f["trace_ms"])
                if world > 1:
                    for k in ("trace_ms", "blur_ms", "halo_ms", "gather_ms", "frame_ms"):
                        if f[k] > 0:
                            diag[k].append(f[k])
            if world > 1:
                diag["enqueue_us"].append(f["enqueue_us"])
                diag["rows"] = f["y1"] - f["y0"]
                diag["cost"] = f["cost"]
                diag["redone"] += int(f["redone"])
            last["f"] = f
        if world == 1:
            # N = 1: the slot ring of the frames API, frames stay on the device
            # --prepare early (default): the tables of frame i are prepared BEFORE the host waits for a free slot.  Their
            # upload (own stream, into a copy of the tables no frame in flight reads) is then done when frame i is
            # submitted, and its trace launch needs no wait between the streams in front of it -- a packet that costs
            # the queue ~5 us per frame (tools/ubench/graph_gap.hip).  Same calls and the same work per frame as
            # --prepare late, which is the reference's order (level_prepare_render right before the trace, main.c:95-107).
            for i in range(n):
                k = i % nres
                if early:
                    r.set_objects(spheres)
                if i >= nres:
                    note(r.wait_frame(k))
                if not early:
                    r.set_objects(spheres)
                r.submit_frame(cam, sec, k)
            for i in range(max(0, n - nres), n):
                note(r.wait_frame(i % nres))
        else:
            # N > 1: tiled_depth[0] frames in flight (three; the sweep also tries five)
            behind = tiled_depth[0] - 1
            for i in range(n):
                if not early or i == 0:
                    r.set_objects(spheres)
                r.tiled_submit(cam, sec)
                if early and i + 1 < n:
                    r.set_objects(spheres)          # frame i+1's tables, before the wait below
                if i >= behind:
                    note(r.tiled_wait())
            for _ in range(min(n, behind)):
                note(r.tiled_wait())

    class LegFailed(RuntimeError):
        pass

    def all_ranks_ok(err):
        """the collective that closes a piece of work on every rank, whatever happened in it locally: True if no rank failed.  Every
        rank makes the SAME sequence of collectives, exception or not -- a rank that raised and went off to some other collective
        would leave the others in a barrier that never completes"""
        if world == 1:
            if err is not None:
                raise err
            return True
        flag = torch.tensor([0.0 if err is not None else 1.0], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return float(flag.item()) == 1.0

    def guarded(fn):
        """fn() on every rank, then the agreement: raises LegFailed on EVERY rank if it raised on any"""
        err = None
        try:
            # test hook (tests/test_gpu_bench_ranks.py): PWN_BENCH_FAIL_LEG=LEG:RANK -- that rank's share of that leg raises
            hook = os.environ.get("PWN_BENCH_FAIL_LEG", "")
            if hook and hook.rsplit(":", 1)[0] == leg_name[0] and int(hook.rsplit(":", 1)[1]) == rank:
                raise RuntimeError("injected failure (PWN_BENCH_FAIL_LEG)")
            fn()
            torch.cuda.synchronize()
        except Exception as e:                                       # noqa: BLE001 -- agreed on below, reported in the line
            err = e
            board.note_error("rank %d: %s: %s" % (rank, leg_name[0], e))
        if not all_ranks_ok(err):
            raise LegFailed(("rank %d: %s" % (rank, err)) if err is not None else "another rank failed (see stage_reached)")

    def block():
        """exactly K steps between barrier + synchronize; seconds, max over ranks"""
        barrier()
        t0 = time.perf_counter()
        guarded(lambda: run(args.steps))
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    leg_name = ["headline"]

    def leg(warmup, min_time, max_blocks=500):
        """warm-up, then K-step blocks until min_time seconds were timed: (median block seconds, all blocks)"""
        if world > 1:
            board.mark("%s: warm-up" % leg_name[0])
        guarded(lambda: run(warmup))
        diag_reset()
        blocks = []
        while True:
            if world > 1:
                board.mark("%s: block %d" % (leg_name[0], len(blocks)))
            blocks.append(block())
            # all ranks see the same (max-reduced) times, so they stop together
            if sum(blocks) >= min_time or len(blocks) >= max_blocks:
                break
        return float(np.median(blocks)), blocks

    if world == 1 and overlap_on:
        # events between the kernels of frames that share the chip would time neither kernel by itself and cost a few
        # microseconds of pipeline each: the headline leg records none, the roofline leg below times solo launches
        r.set_frame_timing(0)
    if world > 1:
        board.mark("headline")
        dog.arm(args.headline_timeout, "headline leg")
        die_here("headline")
    try:
        dt, block_s = leg(args.warmup, args.min_time)
    except Exception as e:                                           # noqa: BLE001
        if world == 1:
            raise
        # a deadline of the library passed on THIS rank (PWN_ETIMEDOUT: a peer stopped answering) or a launch failed: say so
        # on the board and leave; the others find it there when their own deadline, or the launcher's SIGTERM, ends them
        board.note_error("rank %d: %s" % (rank, e))
        dog.bail("rank %d: the headline leg failed: %s" % (rank, e))
    if world > 1:
        board.mark("headline_done")
        dog.disarm()
    room_state = r.trace_room_state()          # PWN_OPT_TRACE_ROOM as the headline leg left it
    roofline_leg = None
    if world == 1 and overlap_on:
        # ---- the roofline's launch duration: the same loop on ONE compute stream, every time_every-th frame between HIP
        # events on the launch stream -- a launch by itself, as `roofline` is defined
        r.set_frame_overlap(False)
        r.set_frame_timing(max(1, args.time_every))
        dt1, bl1 = leg(max(2, args.warmup // 2), min(args.min_time, 1.0))
        roofline_leg = {"what": "the resident loop on one compute stream (PWN_OPT_FRAME_OVERLAP 0), where a launch runs by itself",
                        "ms_per_step": round(dt1 / args.steps * 1e3, 4), "value": round(w * h * args.steps / dt1 / 1e6, 3), "blocks": len(bl1)}
        r.set_frame_overlap(True)
    trace_ms = max_over_ranks(float(np.mean(launch_ms)) if launch_ms else 0.0)
    launch_ms_headline = list(launch_ms)
    ranks = None
    sweep = None
    cuts_now = None
    if world > 1:
        tinfo = r.tiled_info()
        redone = max_over_ranks(float(tinfo["frames_redone"]))
        ranks = per_rank()
        cuts_now = [int(v) for v in r.tiled_get_cuts()[0]]

    # ---- outside the timed region: parity of the last frame, work counters ----
    parity = None
    frame_hash = None
    oracle = None
    if rank == 0 and last["f"] is not None:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        try:
            import oracle  # checker only: FNV of the frame vs the compiled reference's golden
            frame_hash = oracle.fnv64(r.read_plane(last["f"]["d_sbuf"]))
            with open(os.path.join(GOLD, "frames.json")) as f:
                cases = json.load(f)["cases"]
            want = [c for c in cases if c["level"] == args.level and (c["w"], c["h"]) == (w, h)
                    and c["sec"] == 0.0 and c["nspheres"] == len(spheres) and c["name"].startswith(("level_spawn", args.level + "_cam0"))]
            if want and args.level == "pwnfps_level":
                parity = bool(frame_hash == (want[0]["post"] if args.blur else want[0]["pre"]))
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("parity check skipped: %s\n" % e)

    nonfinite = None
    if world == 1 and rank == 0:
        try:
            nonfinite = nonfinite_scenes(pwnfps_amd)
        except Exception as e:  # noqa: BLE001
            nonfinite = {"error": str(e)}
    counters = None
    pcie = None
    kernel_ms = None
    post = {"note": None}

    def build_line():
        nonlocal trace_ms
        pix = w * h
        not_rccl = world > 1 and transport != "rccl"
        if world == 1:
            strip_pix, par = pix, "rows/1, %d frames in flight on %s" % (nres, "two compute streams" if overlap_on else "one compute stream")
        else:
            # the roofline's launch: the rank whose trace launches took longest, with the rows it traced
            tm = [v if v is not None else 0.0 for v in ranks["trace_ms"]]
            slow = int(np.argmax(tm))
            strip_pix = ranks["rows"][slow] * w
            trace_ms = tm[slow]
            par = "rows/%d with moving cuts, two grouped %s send/recv launches per frame in order on the frame's own stream (%s in front of the blur; the gather of the finished strips behind it), 3 frames in flight on %s" % (
                world, transport.upper(), ("%d halo rows per neighbour" % tinfo["halo_rows"]) if tinfo["halo_rows"] else "whole pre-blur strips to every rank",
                "two compute streams" if tinfo["two_streams"] else "one compute stream")
        achieved = TRACE_BYTES_PER_PIXEL * strip_pix / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
        vf = valu_fractions("pwn_trace_kernel", w, h, trace_ms) if world == 1 else None
        line = {
            "metric": ("Mpixels/s at %dx%d (level.txt scene, trace + blur), frames resident on the device; mean steps/ray alongside"
                       % (w, h)) + ("; d2h_inclusive = the rate with every frame delivered to the host (SURVEY 8d)" if not args.no_d2h else "")
                      + (" -- MEASURED OVER THE HOST-STAGED SHARED-MEMORY TEST TRANSPORT, NOT RCCL OVER xGMI" if not_rccl else ""),
            "transport": (None if world == 1 else ("rccl" if transport == "rccl" else "shm: host-staged test transport, NOT RCCL over xGMI")),
            "value": round(pix * args.steps / dt / 1e6, 3),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "timing": {"blocks_of_k_steps": len(block_s), "value_is": "median block",
                       "block_ms_p10_p50_p90": [round(float(np.percentile(block_s, q)) * 1e3, 4) for q in (10, 50, 90)],
                       "first_block_ms": round(block_s[0] * 1e3, 4),
                       "trace_launch_ms_p10_p50_p90": [round(float(np.percentile(launch_ms_headline, q)), 4) for q in (10, 50, 90)] if launch_ms_headline else None,
                       "launches_timed": len(launch_ms_headline),
                       "launches_timed_are": ("every %d-th frame of the roofline leg (HIP events on the launch stream): the same loop on one compute "
                                              "stream, where a launch runs by itself" if roofline_leg else
                                              "every %d-th frame of the timed region (HIP events on the launch stream)") % max(1, args.time_every),
                       "roofline_leg": roofline_leg},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "pwnfps level.txt scene (14 game.lua spheres, spawn pose, sec_current=0), "
                                   "%dx%d, POSTPROC_BLUR=%d, one frame per step" % (w, h, args.blur),
                       "level": args.level, "width": w, "height": h, "blur_passes": args.blur,
                       "parallelism": par,
                       "trace_room": dict(room_state, what="PWN_OPT_TRACE_ROOM: workgroups the persistent trace grid leaves free for the other stream's kernels; option -1 = the library compares 0 with one per CU on windows of delivered frames and keeps the faster"),
                       "host_loop": ("set_objects(i) / wait for a free slot / submit(i)" if args.prepare == "early"
                                     else "wait for a free slot / set_objects(i) / submit(i)") + ", every frame re-bins and re-uploads the spheres",
                       "frames_repeated_with_whole_strips": int(redone) if world > 1 else 0},
            "roofline": {"bound": "hbm", "kernel": "pwn_trace_kernel", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": pmc_traffic("pwn_trace_kernel", w, h) if world == 1 else None,
                         "traffic_unit": "bytes/launch (PMC WRITE_SIZE + 2*FETCH_SIZE, profiles/pmc_latest.csv)",
                         "algorithmic_bytes_per_launch": TRACE_BYTES_PER_PIXEL * strip_pix,
                         "instruction_issue": pmc_issue_rate("pwn_trace_kernel", w, h, trace_ms) if world == 1 and trace_ms > 0 else None,
                         "bytes_per_pixel": TRACE_BYTES_PER_PIXEL, "pixels_per_launch": strip_pix,
                         "avg_launch_ms": round(trace_ms, 4),
                         # what really bounds the kernel, against the ARCHITECTURE: VALU instructions issued over what 1024 SIMDs can
                         # issue in the launch's time at 1.2 per ns, and the same weighted by the lanes that were active
                         "valu_issue_frac_of_peak": vf["valu_issue_frac_of_peak"] if vf else None,
                         "lane_slot_frac": vf["lane_slot_frac"] if vf else None,
                         "valu": vf,
                         # ... and against the builder's own cost model (profiles/r5_issue_model.json: per-block instruction mix x
                         # measured execution counts x issue cost per opcode class from this repo's microbenchmark) -- a ratio, not a
                         # roofline fraction (it was `issue_frac` until round 3)
                         "model_over_measured": model_over_measured(w, h, args.level, trace_ms) if world == 1 else None,
                         "note": "VALU-issue-bound DDA: tables live in LDS, compulsory HBM traffic is the 8 B/pixel written; read "
                                 "valu_issue_frac_of_peak / lane_slot_frac for how far the kernel is from the chip's VALU rate"},
            # the second kernel of a frame, the one that really is a memory gather: 12 algorithmic
            # bytes per pixel (read colour 4 + depth 4, write 4), single-GPU figure from HIP events
            "blur_roofline": ({"bound": "hbm", "kernel": "pwn_blur_tiled_kernel", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                               "achieved": round(12 * pix / (kernel_ms["blur"] * 1e-3) / 1e9, 3),
                               "frac": round(12 * pix / (kernel_ms["blur"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                               "bytes_per_pixel": 12, "avg_launch_ms": kernel_ms["blur"]}
                              if kernel_ms and kernel_ms.get("blur", 0) > 0 else None),
            "frame_bytes_per_pixel": FRAME_BYTES_PER_PIXEL,
            "frame_gbs": round(FRAME_BYTES_PER_PIXEL * pix * args.steps / dt / 1e9, 3),
            "parity_vs_reference_golden": parity,
            "frame_fnv64": frame_hash,
            "parity": {"headline_frame_equals_reference_golden": parity,
                       "golden_pinned_by": "oracle/_ref/libpwnref_tab.so (the reference's headers, its flags, rcpps/rsqrtps from captured tables) and, "
                                           "for all 29 golden cases, by libpwnref_hw.so with the Intel host's own instructions (frames.json hw_equal)",
                       "contract": "bit-exact, no tolerance: colour (four bytes), depth (fp32 bits) and the work counters.  Where the reference itself "
                                   "divides by zero (a ramp whose tilt cancels ray.y, trace.h:461) its -ffinite-math-only build is not defined: "
                                   "there the frames equal the same sources built with -fno-finite-math-only everywhere, and the shipped-flags "
                                   "build wherever depth is finite (DESIGN.md 2)",
                       "nonfinite_scenes": nonfinite},
        }
        if world > 1:
            line["tiling"] = {k: tinfo[k] for k in ("rows_per_rank", "halo_rows", "groups", "frames", "frames_redone", "bytes_sent", "bytes_received",
                                                    "max_rows", "balance_every", "grid_reserve", "two_streams", "recuts")}
            line["tiling"]["transport"] = transport
            line["tiling"]["choreography"] = ("in-stream: trace, halo rows, blur, gather and words of a frame in order on the frame's own compute stream, frames "
                                              "alternating between two (PWN_OPT_TILED_CHOREO; sweep.choreo_split = the other form)")
            line["tiling"]["preflight"] = preflight
            line["tiling"]["preflight_summary"] = preflight_summary
            line["tiling"]["rccl_nonblocking"] = tinfo.get("rccl_nonblocking")
            line["tiling"]["communicators"] = tinfo.get("communicators")
            line["tiling"]["deadlines_s"] = {"library_init": lib_init_s, "library_wait": lib_wait_s, "bring_up": args.bringup_timeout,
                                             "headline": args.headline_timeout, "post": args.post_timeout}
            if transport_note:
                line["tiling"]["transport_note"] = transport_note
            line["tiling"]["cuts"] = cuts_now
            # every rank's own account of the headline leg: kernel times of the timed frames (HIP events; with two compute
            # streams a trace shares the chip with the neighbour frames' kernels), the two grouped exchanges (halo = this
            # frame's border rows, gather = its finished strips + the ranks' words), host time inside pwn_tiled_submit per frame, the rows of its strip and what they cost
            line["tiling"]["per_rank"] = ranks
            if first_legs is not None:
                # the forms measured with the headline's own blocks; `value` is the default's (the contract: the configuration the metric
                # names), `best` the fastest of them on this node; predicted_ms = DESIGN.md 6's link model on this run's kernel times
                line["tiling"]["first_legs"] = first_legs
                ok_legs = {k: v for k, v in first_legs.items() if v.get("value")}
                if ok_legs:
                    bk = max(ok_legs, key=lambda k: ok_legs[k]["value"])
                    line["best"] = {"leg": bk, "value": ok_legs[bk]["value"], "unit": "Mpixels/s", "measured_ms": ok_legs[bk]["measured_ms"],
                                    "predicted_ms": ok_legs[bk].get("predicted_ms"),
                                    "note": "host_sink delivers every frame to the host; the others leave it on a device"}
            if sweep_dead[0] is not None:
                line["tiling"]["sweep_stopped_by"] = sweep_dead[0]
            if sweep is not None:
                line["tiling"]["sweep"] = sweep
            if post["note"]:
                line["tiling"]["post_note"] = post["note"]
        if counters:
            line["work"] = counters
        if kernel_ms:
            line["kernel_ms"] = kernel_ms
        if pcie:
            line["d2h_inclusive"] = pcie
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        # `vs_baseline` stays null: BASELINE.md holds no published number for this metric.  The ratio a reader wants -- the
        # metric as SURVEY 8(d) words it (every frame delivered to the host) over the reference's own code on this box's
        # cores in the same run -- is here:
        if cpu_line is not None and pcie and pcie.get("value"):
            line["vs_cpu_baseline"] = {"value": round(pcie["value"] / cpu_line["value"], 1),
                                       "value_vs_1_thread": round(pcie["value"] / cpu_line["value_1_thread"], 1) if cpu_line.get("value_1_thread") else None,
                                       "what": "d2h_inclusive.value / cpu_baseline.value: frames delivered to the host over PCIe against the reference CPU "
                                               "path on %d host threads; a reported ratio, not the optimisation target" % cpu_line["cores"]}
        return line

    # ---- N > 1: what follows the headline (sweep, host-sink leg) runs under a deadline on every rank: a leg that hangs -- a transport
    # that does not come up a second time, a peer that died -- must not cost the headline number.  When it passes, rank 0 prints the
    # line with what it has ("incomplete": true, the stages) and every rank leaves with status 3 (the process's exit takes its GPU
    # queues down; ADVICE r3: not status 0, a driver that keys on the exit code must not take a wedged run for a clean one).
    if world > 1:
        def line_so_far():
            post["note"] = "the legs after the headline did not finish (--post-timeout %.0f s): sweep / d2h_inclusive hold what was measured until then" % args.post_timeout
            return build_line()
        partial["build"] = line_so_far
        if args.post_timeout > 0:
            dog.arm(args.post_timeout, "the legs after the headline")

    # ---- N > 1: the forms that the link model expects to matter most, with the headline's own K-step blocks, BEFORE
    # the long sweep: `value` stays the default's; `best` says which of them a host should pick on this node.
    tiling_up = world > 1
    sweep_dead = [None]                  # a leg failed on some rank: what it said; nothing that needs the tiling runs after it

    def link_model(form):
        """what the form should cost per frame by the link rates DESIGN.md 6 assumes (60 GB/s per direction and xGMI link, 52 GB/s per
        PCIe link), from THIS run's kernel times: max(kernels of the slowest rank, the exchange that cannot hide)"""
        if ranks is None:
            return None
        k = max((t or 0.0) + (b or 0.0) for t, b in zip(ranks["trace_ms"], ranks["blur_ms"]))
        strip = 4.0 * w * h / world
        halo = 2.0 * (tinfo["halo_rows"] * w * 4.0 if tinfo["halo_rows"] else strip * (world - 1)) / 60e9 * 1e3
        if form == "fixed_root":
            x = halo + strip / 60e9 * 1e3                      # every link into rank 0 carries one strip per frame
        elif form == "rotating_root":
            x = halo + strip / 60e9 * 1e3 / min(2.0, world)    # successive frames' gathers go to different ranks, two in flight
        elif form == "host_sink":
            x = max(halo, strip / 52e9 * 1e3)                  # every rank's strip over its own PCIe link, beside the halo rows
        else:
            return None
        return round(max(k, x), 4)
    first_legs = None
    if world > 1:
        first_legs = {"default": {"what": "fixed root (rank 0), in-stream choreography: the headline", "measured_ms": round(dt / args.steps * 1e3, 4),
                                  "value": round(w * h * args.steps / dt / 1e6, 3), "predicted_ms": link_model("fixed_root")}}
        leg_name[0] = "first.rotating_root"
        try:
            r.tiled_gather_root(True)
            d2, _bl = leg(max(2, args.warmup // 2), min(args.min_time, 1.0))
            first_legs["rotating_root"] = {"what": "pwn_tiled_gather_root(ROTATE): frame f gathered on rank f mod N", "measured_ms": round(d2 / args.steps * 1e3, 4),
                                           "value": round(w * h * args.steps / d2 / 1e6, 3), "predicted_ms": link_model("rotating_root")}
            r.tiled_gather_root(False)
        except LegFailed as e:
            first_legs["rotating_root"] = {"error": str(e)}
            sweep_dead[0] = "first.rotating_root: %s" % e
    if world > 1 and not args.no_d2h and sweep_dead[0] is None:
        leg_name[0] = "host_sink"
        pcie = host_sink_leg(r, args, w, h, cam, sec, spheres, rank, world, transport, barrier, max_over_ranks,
                             (lambda buf: oracle.fnv64(buf) == frame_hash) if (rank == 0 and oracle is not None and frame_hash is not None) else None,
                             mark=board.mark)
        tiling_up = False                 # (the leg leaves the tiling shut down)
        if pcie and pcie.get("value"):
            first_legs["host_sink"] = {"what": "pwn_tiled_host_sink: every rank's strip over its own PCIe link into one host frame, no gather",
                                       "measured_ms": pcie["ms_per_step"], "value": pcie["value"], "predicted_ms": link_model("host_sink")}
        elif pcie:
            first_legs["host_sink"] = {"error": pcie.get("error")}
    # ---- N > 1: the same run, other settings, a fraction of a second each: what a first multi-GPU run should look at
    if world > 1 and args.sweep_time > 0 and not tiling_up and sweep_dead[0] is None:
        tiling_up, _ = bring_up(transport, tag="sweep:")
    if world > 1 and args.sweep_time > 0 and tiling_up and sweep_dead[0] is None:
        sweep = {}

        def point(name, what):
            # (a leg that fails -- a communicator that does not come up a second time, a deadline of the library -- is recorded
            # and ends the sweep on EVERY rank at the same collective; the line is printed whole, the exit status stays 0)
            if sweep_dead[0] is not None:
                sweep[name] = {"skipped": "an earlier leg failed: " + sweep_dead[0]}
                return
            leg_name[0] = "sweep." + name
            try:
                d2, bl = leg(max(2, args.warmup // 2), args.sweep_time, 60)
            except LegFailed as e:
                sweep[name] = {"what": what, "error": str(e)}
                sweep_dead[0] = "%s: %s" % (name, e)
                return
            pr = per_rank()
            sweep[name] = {"what": what, "value": round(w * h * args.steps / d2 / 1e6, 3), "ms_per_step": round(d2 / args.steps * 1e3, 4),
                           "blocks": len(bl), "trace_ms": pr["trace_ms"], "blur_ms": pr["blur_ms"], "halo_ms": pr["halo_ms"],
                           "gather_ms": pr["gather_ms"], "enqueue_us": pr["enqueue_us"], "rows": pr["rows"]}
        def alive(fn, *a):
            """a setter of the tiling between two legs: not on a tiling that a failed leg left in an unknown state"""
            if sweep_dead[0] is None:
                fn(*a)
        reserve0 = tinfo["grid_reserve"]
        for rsv in (0, 16, 64):
            alive(r.tiled_set_reserve, rsv)
            point("reserve_%d" % rsv, "PWN_TILED_RESERVE = %d workgroups of the persistent trace grid left free for the transport's kernels" % rsv)
        alive(r.tiled_set_reserve, reserve0)
        # the host collecting frames four behind its submissions instead of two
        tiled_depth[0] = 5
        point("five_in_flight", "five frames in flight instead of three (PWN_TILED_SLOTS - 1): the host waits for frame f-4 after submitting f")
        tiled_depth[0] = 3
        # the gather spread over the ranks: frame f assembled on rank f mod N (pwn_tiled_gather_root) -- no rank's links carry every
        # frame; what a consumer on every GPU would get
        try:
            alive(r.tiled_gather_root, True)
            point("rotating_root", "pwn_tiled_gather_root(ROTATE): frame f is gathered on rank f mod N in turn instead of always on rank 0")
        except Exception as e:                                       # noqa: BLE001
            sweep["rotating_root"] = {"error": str(e)}
        finally:
            alive(r.tiled_gather_root, False)
        bal = tinfo["balance_every"]
        alive(r.tiled_balance, 0)
        alive(r.tiled_set_cuts, [min(k * tinfo["rows_per_rank"], h) for k in range(world)] + [h])
        point("equal_strips", "pwn_tiled_balance(0) with the equal split (the reference's static schedule, screen.h:63-64)")
        # rank 0 takes in every other rank's finished strip and sends none: with a taller strip of its own the strips that cross the
        # links into it get shorter (the gather is what the prediction says binds)
        tall = min(tinfo["max_rows"], (int(tinfo["rows_per_rank"] * 1.4) + 7) // 8 * 8)
        rest = h - tall
        if world > 2 and rest // (world - 1) >= max(tinfo["halo_rows"], 16):
            each = rest // (world - 1) // 8 * 8
            cuts_tall = [0] + [tall + k * each for k in range(world - 1)] + [h]
            try:
                alive(r.tiled_set_cuts, cuts_tall)
                point("rank0_tall", "rank 0 traces %d rows, the others %d: less crosses the links into rank 0 per frame" % (tall, each))
            except Exception as e:                                   # noqa: BLE001 -- (cuts outside the library's bounds: no point)
                sweep["rank0_tall"] = {"error": str(e)}
        alive(r.tiled_balance, bal)
        # ---- legs that set the tiling up again, the ones a first multi-GPU run learns most from first.  (The options are read by
        # pwn_tiled_init and refuse to change while a tiling exists: each leg shuts the last one down, then sets its own.)
        t_sweep = time.perf_counter()

        def again(name, what, comms=False, streams=2, split=False, overlap=True, halo=None, depth=3, rotate=False):
            # every one of these legs makes its communicator(s) anew: together they stay inside --sweep-budget seconds (every rank
            # takes the same decision: the slowest rank's clock)
            if sweep_dead[0] is not None:
                sweep[name] = {"skipped": "an earlier leg failed: " + sweep_dead[0]}
                return
            if max_over_ranks(time.perf_counter() - t_sweep) > args.sweep_budget:
                sweep[name] = {"skipped": "--sweep-budget %.0f s used up by the legs in front" % args.sweep_budget}
                return
            barrier()
            r.tiled_shutdown()
            r.set_tiled_comms(comms)
            r.set_tiled_streams(streams)
            r.set_tiled_choreo(split)
            r.set_frame_overlap(overlap)
            ok, why = bring_up(transport, halo=halo, tag="sweep.%s:" % name)
            if not ok:
                sweep[name] = {"what": what, "error": "the tiling did not come up: %s" % (why or "on another rank")}
                sweep_dead[0] = "%s: the tiling did not come up" % name
            if ok:
                if rotate:
                    r.tiled_gather_root(True)                        # (gone with the tiling at the next leg's shutdown)
                tiled_depth[0] = depth
                try:
                    point(name, what)
                finally:
                    tiled_depth[0] = 3
        # the other choreography first among the legs that set the tiling up again: the in-stream default has run between GPUs as little as
        # this one -- the first multi-GPU line says which a host should ask for
        again("choreo_split", "PWN_OPT_TILED_CHOREO = split (rounds 2-3): the exchanges on a third stream tied to the kernels by events, blur f enqueued by submit f+1 "
                              "and its gather by submit f+2 (the headline: everything of a frame in order on the frame's own stream)", split=True)
        if first_legs is not None and "value" in sweep.get("choreo_split", {}):
            first_legs["choreo_split"] = {"what": "the split choreography (PWN_OPT_TILED_CHOREO), fixed root", "measured_ms": sweep["choreo_split"]["ms_per_step"],
                                          "value": sweep["choreo_split"]["value"], "predicted_ms": link_model("fixed_root")}
        if transport == "rccl" or os.environ.get("PWN_BENCH_ALL_LEGS"):        # (the test hook runs these lines over shm, where the option changes nothing)
            # a communicator per compute stream: the streams' exchanges do not wait for each other (one communicator runs its
            # launches in the order they were made); then that with three streams and four frames in flight -- on one GPU, a
            # rank exchanging with itself, the fastest form measured
            again("comm_per_stream", "PWN_OPT_TILED_COMMS = per stream: one RCCL communicator per compute stream instead of one for both", comms=True)
            again("comm_per_stream_rotating_root", "a communicator per stream and frame f gathered on rank f mod N: successive frames' gathers run in "
                  "opposite directions over a link and, on communicators of their own, at the same time", comms=True, rotate=True)
            again("three_streams_comm_per_stream", "three compute streams, a communicator each, four frames in flight", comms=True, streams=3, depth=4)
            again("three_streams_comm_per_stream_rotating_root", "the same with frame f gathered on rank f mod N: no rank's links carry every frame and the "
                  "streams' exchanges are independent -- the form with the most overlap", comms=True, streams=3, depth=4, rotate=True)
        again("three_streams", "PWN_OPT_TILED_STREAMS = 3: frame f on compute stream f mod 3 (one GPU: a strip-sized frame 8 % faster, a whole 4K frame 8 % slower)", streams=3)
        again("one_stream", "PWN_OPT_FRAME_OVERLAP 0: every frame's kernels on ONE compute stream", overlap=False)
        again("whole_strips", "halo 0: every rank's whole pre-blur strip to every rank instead of the bounded halo rows", halo=0)
        if transport == "rccl" and args.sweep_nonblocking:
            # (opt-in: an optional leg on the least-travelled path must not be able to cost a run its clean exit)
            # the other way of driving the communicator: non-blocking, every call polled against the deadline (a grouped launch is
            # then handed to a thread of RCCL's and the host waits for it) -- what that costs the host per frame (enqueue_us)
            barrier()
            r.tiled_shutdown()
            os.environ["PWN_TILED_RCCL_MODE"] = "nonblocking"          # (read by pwn_tiled_init; the same on every rank)
            ok3, _ = bring_up(transport, tag="sweep.rccl_nonblocking:")
            if ok3:
                point("rccl_nonblocking", "PWN_TILED_RCCL_MODE=nonblocking: ncclCommInitRankConfig(blocking = 0), every call polled with ncclCommGetAsyncError")
            del os.environ["PWN_TILED_RCCL_MODE"]

    if world == 1:
        r.set_call_strips(0)               # counters, wave stamps and kernel_ms: one launch per pass
        r.set_counters(True)
        sb = np.empty((h, w), np.uint32)
        r.trace_screen_centred(cam, sec, want_z=False, sbuf=sb)
        st = r.stats()
        r.set_counters(False)
        counters = {"rays_per_pixel": round(st["rays"] / (w * h), 4),
                    "steps_per_ray": round(st["steps"] / max(st["rays"], 1), 4),
                    "portal_crossings_per_ray": round(st["portals"] / max(st["rays"], 1), 4),
                    "sphere_tests_per_ray": round(st["sphere_tests"] / max(st["rays"], 1), 4),
                    # lanes doing a cell step / lanes of the wave64s running the walk loop
                    "walk_active_lane_fraction": round(st["steps"] / max(64 * st["wave_steps"], 1), 4)}
        # share of the kernel's duration the average wave64 is resident: start / end stamps of every
        # wave (constant 100 MHz clock) in an UNcounted frame of the timed build
        r.set_wave_log(True)
        r.trace_screen_centred(cam, sec, want_z=False, sbuf=sb)
        sw = r.stats()
        r.set_wave_log(False)
        counters["mean_wave_residency"] = round(sw["wave_time"] / max(sw["waves"] * sw["kernel_span"], 1), 4)
        counters["trace_kernel_span_ms"] = round(sw["kernel_span"] / 1e5, 4)
        counters["waves"] = sw["waves"]
        # ---- the one call an unchanged reference loop makes per frame (main.c:107): wall time of pwn_trace_screen_centred incl. the
        # hand-over into sbuf, SURVEY 8(d)'s literal metric.  In one piece (trace, blur, then 33 MB over PCIe: what rounds 1-4
        # reported), and in row strips with the copies beside the kernels (PWN_OPT_CALL_STRIPS, the default from 3 Mpixels on) into
        # the host's buffer as malloc'ed and as registered once (pwn_host_register: INTEGRATION.md's line behind main.c:395-400)
        def blocking(n=7):
            best = 1e9
            for _ in range(n):
                t1 = time.perf_counter()
                r.trace_screen_centred(cam, sec, want_z=False, sbuf=sb)
                best = min(best, time.perf_counter() - t1)
            return best
        blocking_one_piece = blocking()
        st = r.stats()                     # kernel times of an uncounted frame, one launch per pass
        kernel_ms = {"trace": round(st["trace_ms"], 4), "blur": round(st["blur_ms"], 4)}
        blocking_best = blocking_one_piece
        if not args.no_d2h:
            # (--no-d2h is the form run under rocprofv3: no strip-sized launches, so the profile's per-kernel average is the
            # whole-frame launch the roofline object quotes)
            r.set_call_strips(-1)
            blocking_pageable = blocking()
            strips_pageable = r.call_strips_state()["strips_last"]
            r.host_register(sb)
            # (a context's first sixteen calls in strips find out whether its chunks go out on one copy stream or on two: the steady
            # state is what is timed)
            for _ in range(20):
                r.trace_screen_centred(cam, sec, want_z=False, sbuf=sb)
            blocking_best = blocking(9)
            cs_state = r.call_strips_state()
            blocking_same = bool(oracle.fnv64(sb) == frame_hash) if (oracle is not None and frame_hash is not None) else None
            r.host_unregister(sb)
            blocking_call = {"one_piece_mpix_s": round(w * h / blocking_one_piece / 1e6, 2), "one_piece_ms": round(blocking_one_piece * 1e3, 4),
                             "strips_pageable_mpix_s": round(w * h / blocking_pageable / 1e6, 2), "strips_pageable_ms": round(blocking_pageable * 1e3, 4),
                             "strips_registered_mpix_s": round(w * h / blocking_best / 1e6, 2), "strips_registered_ms": round(blocking_best * 1e3, 4),
                             "strips": cs_state["strips_last"], "strips_pageable": strips_pageable, "blur_repeated": cs_state["redone"],
                             "copy_streams": cs_state["copy_streams"], "reach_depth": cs_state["reach_depth"],
                             "frame_equals_resident_frame": blocking_same,
                             "pcie_floor_ms": round(4 * w * h / 54e9 * 1e3, 4),
                             "what": "best of 7 (registered: of 9 behind 20 calls) wall times of one pwn_trace_screen_centred(cam, sec, sbuf, NULL): one launch per pass and then "
                                     "the copy (PWN_OPT_CALL_STRIPS 0); in row strips (the default) into a malloc'ed sbuf; in row strips into the same "
                                     "sbuf registered with pwn_host_register"}
        r.set_call_strips(-1)
    if world == 1 and not args.no_d2h:
        pcie = d2h_leg_one_gpu(r, args, w, h, cam, sec, spheres, blocking_best,
                               (lambda buf: oracle.fnv64(buf) == frame_hash) if (oracle is not None and frame_hash is not None) else None)
        pcie["blocking_call"] = blocking_call
    dog.disarm()
    post["note"] = None
    if rank == 0:
        print(json.dumps(build_line()), file=line_out, flush=True)

    if world > 1:
        barrier()
        r.tiled_shutdown()
    r.close()
    if world > 1:
        dist.destroy_process_group()


_RUN = {"dog": None, "board": None, "world": 1, "rank": 0}

if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:                                       # noqa: BLE001
        # N > 1: an exception on one rank (a control-plane collective that lost its peer, a launch that failed) must
        # not end in a traceback alone: the same diagnostic line, the same exit status as a deadline
        if _RUN["dog"] is not None and _RUN["world"] > 1:
            import traceback
            traceback.print_exc()
            _RUN["board"].note_error("rank %d: %s: %s" % (_RUN["rank"], type(e).__name__, e))
            _RUN["dog"].bail("rank %d: %s: %s" % (_RUN["rank"], type(e).__name__, str(e)[:500]))
        raise
