"""bench.py with 2 and 3 ranks, the way the driver starts it (torch.distributed.run, one process per
rank), on the one GPU of the test box: PWN_BENCH_ONE_DEVICE=1 puts every rank on device 0 and
PWN_BENCH_TRANSPORT=shm replaces RCCL (which cannot run two ranks on one device) by the shared
memory transport; process group, id broadcast, pwn_tiled_* with HIP strip kernels, frames in
flight and the JSON line are the real thing.  The last frame of the timed loop must be the
compiled reference's golden frame."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("members,size,golden", [(2, (3840, 2160), "a95833dac9624326"), (4, (1280, 720), "078fb94a5cd068f5")])
def test_bench_single_process_over_one_device(members, size, golden):
    """bench.py --gpus N --single-process: ONE process, one handle (pwn_init_multi), the library's member threads; here with every
    member on device 0 (PWN_BENCH_ONE_DEVICE).  The line's keys, and the golden frame from the resident loop, the delivered
    loop and the blocking call."""
    env = dict(os.environ)
    env.update(PWN_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(members), "--single-process", "--steps", "6", "--warmup", "2", "--min-time", "0.2",
           "--time-every", "2", "--width", str(size[0]), "--height", str(size[1])]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == members and d["single_process"] is True and d["transport"] == "local" and d["devices"] == [0] * members
    assert d["steps"] == 6 and d["value"] > 0 and d["scaling"] == "strong" and d["unit"] == "Mpixels/s"
    assert d["frame_fnv64"] == golden
    if size == (3840, 2160):
        assert d["parity_vs_reference_golden"] is True
    assert "ONE DEVICE" in d["metric"]
    t = d["tiling"]
    assert t["cuts"][0] == 0 and t["cuts"][-1] == size[1] and len(t["cuts"]) == members + 1 and t["frames_redone"] == 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] == "pwn_trace_kernel" and r["achieved"] > 0 and r["pixels_per_launch"] <= size[0] * size[1]
    q = d["d2h_inclusive"]
    assert q["value"] > 0 and q["last_frame_equals_resident_frame"] is True
    assert q["blocking_call_mpix_s"] > 0 and q["blocking_call_frame_equals_resident_frame"] is True
    # under torch.distributed.run it refuses: it IS the one process
    cmd2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
            os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process"]
    if members == 2:
        p2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert p2.returncode != 0 and "one process" in (p2.stderr + p2.stdout)


@pytest.mark.parametrize("world,size,golden", [(2, (3840, 2160), "a95833dac9624326"), (3, (1280, 720), "078fb94a5cd068f5")])
def test_bench_with_ranks_on_one_device(world, size, golden):
    env = dict(os.environ)
    env.update(PWN_BENCH_ONE_DEVICE="1", PWN_BENCH_TRANSPORT="shm", MASTER_ADDR="127.0.0.1")
    if world == 3:
        env.update(PWN_BENCH_ALL_LEGS="1")          # the sweep legs that only an RCCL run has, so that their lines have run somewhere
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2", "--min-time", "0.2", "--sweep-time", "0.05",
           "--time-every", "2", "--width", str(size[0]), "--height", str(size[1])]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    # ONE line on stdout and nothing else (gloo announces its connections on descriptor 1: bench.py sends that to stderr)
    assert [l for l in p.stdout.splitlines() if l.strip()] == lines, p.stdout[:600]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 6 and d["value"] > 0
    assert d["frame_fnv64"] == golden
    if size == (3840, 2160):
        assert d["parity_vs_reference_golden"] is True
    t = d["tiling"]
    assert t["transport"] == "shm" and t["frames_redone"] == 0 and t["halo_rows"] > 0
    assert d["config"]["frames_repeated_with_whole_strips"] == 0
    # a number measured over the test transport must be impossible to take for RCCL over xGMI
    assert d["transport"].startswith("shm") and "NOT RCCL" in d["transport"] and "NOT RCCL" in d["metric"]
    # what a first multi-GPU run wants on record before it starts, per rank; how the communicator is driven; the deadlines
    pf = t["preflight"]
    assert [q["rank"] for q in pf] == list(range(world))
    for q in pf:
        assert q["devices_visible"] >= 1 and q["can_access_peer"][q["device"]] == 1 and q["rccl_version"] > 20000 and os.path.basename(q["librccl"]).startswith("librccl")
        assert q["rccl_mode"] == "blocking" and q["init_timeout_ms"] == int(t["deadlines_s"]["library_init"] * 1000)
    assert t["preflight_summary"]["every_device_reaches_every_other"] is True and len(t["preflight_summary"]["librccl_files"]) == 1
    assert t["deadlines_s"]["bring_up"] == 300.0 and t["deadlines_s"]["library_init"] < t["deadlines_s"]["bring_up"]
    assert "incomplete" not in d and "stage_reached" not in d
    # every rank's own account of the headline leg, and the cuts (they move with what the strips cost)
    pr = t["per_rank"]
    for k in ("trace_ms", "blur_ms", "halo_ms", "gather_ms", "frame_ms", "enqueue_us", "rows", "cost", "frames_redone", "timed_frames", "trace_room"):
        assert len(pr[k]) == world, k
    assert sum(pr["rows"]) == size[1] and all(c > 0 for c in pr["cost"]) and all(v > 0 for v in pr["trace_ms"]) and all(v > 0 for v in pr["enqueue_us"])
    # (halo_ms was read from an event without a time stamp and never filled)
    assert all(v is not None and v > 0 for v in pr["halo_ms"]) and all(v is not None and v > 0 for v in pr["blur_ms"]), pr
    assert len(t["cuts"]) == world + 1 and t["cuts"][0] == 0 and t["cuts"][-1] == size[1]
    assert t["balance_every"] == 8 and t["two_streams"] == 1 and t["max_rows"] >= t["rows_per_rank"] and t["grid_reserve"] == 0
    assert d["roofline"]["pixels_per_launch"] in [rows * size[0] for rows in pr["rows"]]
    # the in-run sweep: room left for the transport's kernels, equal strips, one compute stream, whole strips, the other choreography
    sw = t["sweep"]
    rccl_legs = {"comm_per_stream", "comm_per_stream_rotating_root", "three_streams_comm_per_stream", "three_streams_comm_per_stream_rotating_root"}
    assert set(sw) - {"rank0_tall"} - rccl_legs == {"reserve_0", "reserve_16", "reserve_64", "rotating_root", "equal_strips", "one_stream", "whole_strips", "choreo_split", "three_streams", "five_in_flight"}
    assert (set(sw) & rccl_legs) == (rccl_legs if world == 3 else set())
    assert t["choreography"].startswith("in-stream") and all(q["choreography"] == "instream" for q in pf)
    assert ("rank0_tall" in sw) == (world > 2)
    for name, pt in sw.items():
        assert pt["value"] > 0 and pt["ms_per_step"] > 0 and len(pt["trace_ms"]) == world and len(pt["rows"]) == world, name
    assert sw["equal_strips"]["rows"] == [min((k + 1) * t["rows_per_rank"], size[1]) - min(k * t["rows_per_rank"], size[1]) for k in range(world)]
    # the forms a first multi-GPU run decides between, measured with the headline's blocks in front of the sweep, each beside the
    # link model's prediction; `value` stays the default's, `best` names the fastest
    fl = t["first_legs"]
    assert set(fl) == {"default", "rotating_root", "host_sink", "choreo_split"}
    for name, leg in fl.items():
        assert leg["value"] > 0 and leg["measured_ms"] > 0 and leg["predicted_ms"] > 0, (name, leg)
    assert fl["default"]["value"] == d["value"] and fl["host_sink"]["value"] == d["d2h_inclusive"]["value"]
    assert d["best"]["leg"] in fl and d["best"]["value"] == max(v["value"] for v in fl.values()) and d["best"]["value"] >= d["value"]
    assert "sweep_stopped_by" not in t
    # the host-delivered leg (pwn_tiled_host_sink): every rank's strip into one shared frame, hashed against the resident one
    hs = d["d2h_inclusive"]
    assert hs["value"] > 0 and hs["pcie_links"] == world and hs["last_frame_equals_resident_frame"] is True
    assert 0 < hs["bytes_over_pcie_per_frame_and_rank"] <= t["max_rows"] * size[0] * 4


def test_bench_line_survives_a_leg_that_does_not_finish():
    """The legs after the headline (sweep, host-sink leg) run under a deadline: when it passes -- here a sweep made far too long
    for it -- rank 0 prints the line with the headline figures, "incomplete": true, the stage every rank was in and a note, and
    every rank leaves with a NON-ZERO status (a driver that keys on the exit code must not take it for a clean run)."""
    env = dict(os.environ)
    env.update(PWN_BENCH_ONE_DEVICE="1", PWN_BENCH_TRANSPORT="shm", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--min-time", "0.1", "--sweep-time", "30",
           "--post-timeout", "0.2", "--time-every", "2", "--width", "1280", "--height", "720"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["frame_fnv64"] == "078fb94a5cd068f5"
    assert d["incomplete"] is True and "legs after the headline" in d["error"]
    # (the host-delivered leg comes first after the headline, then the sweep: whichever the deadline caught them in)
    assert [s["rank"] for s in d["stage_reached"]] == [0, 1] and all(s["stage"].startswith(("sweep", "host_sink", "headline_done", "first")) for s in d["stage_reached"]), d["stage_reached"]
    assert "did not finish" in d["tiling"]["post_note"] and len(d["tiling"]["per_rank"]["trace_ms"]) == 2


def test_a_sweep_leg_that_fails_is_recorded_and_the_line_is_complete():
    """a PWN_ETIMEDOUT or a launch error in an optional leg used to raise out of main() and turn a run whose headline was
    measured into `incomplete: true`, exit 3.  Now the leg's failure is agreed on by all ranks at the collective that closes the
    leg, recorded, the legs behind it are skipped, and the line comes out whole with status 0."""
    env = dict(os.environ)
    env.update(PWN_BENCH_ONE_DEVICE="1", PWN_BENCH_TRANSPORT="shm", MASTER_ADDR="127.0.0.1", PWN_BENCH_FAIL_LEG="sweep.five_in_flight:1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--min-time", "0.1", "--sweep-time", "0.05",
           "--headline-timeout", "16", "--time-every", "2", "--width", "1280", "--height", "720"]          # (the library's wait deadline: a quarter of it)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    sw = d["tiling"]["sweep"]
    assert "incomplete" not in d and d["value"] > 0 and d["frame_fnv64"] == "078fb94a5cd068f5"
    # (rank 1's share of the leg raised; rank 0, whose line this is, ran into the library's deadline waiting for it -- either text)
    err = sw["five_in_flight"]["error"]
    assert sw["reserve_64"]["value"] > 0 and ("injected failure" in err or "did not answer" in err or "another rank" in err) and "value" not in sw["five_in_flight"]
    assert d["tiling"]["sweep_stopped_by"].startswith("five_in_flight")
    for name in ("equal_strips", "choreo_split", "one_stream", "whole_strips"):
        assert "skipped" in sw[name], (name, sw[name])
    assert d["d2h_inclusive"]["value"] > 0 and d["tiling"]["first_legs"]["rotating_root"]["value"] > 0


def test_sweep_legs_that_set_the_tiling_up_again_stay_inside_their_budget():
    """--sweep-budget: the legs that make their communicators anew are skipped, on every rank alike, once the ones in front have used
    the budget up -- the line is complete and the run leaves with status 0."""
    env = dict(os.environ)
    env.update(PWN_BENCH_ONE_DEVICE="1", PWN_BENCH_TRANSPORT="shm", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--min-time", "0.1", "--sweep-time", "0.05",
           "--sweep-budget", "0", "--no-d2h", "--time-every", "2", "--width", "1280", "--height", "720"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    sw = d["tiling"]["sweep"]
    assert "incomplete" not in d and d["value"] > 0 and sw["reserve_16"]["value"] > 0 and sw["five_in_flight"]["value"] > 0
    for name in ("three_streams", "choreo_split", "one_stream", "whole_strips"):
        assert "skipped" in sw[name] and "value" not in sw[name], (name, sw[name])


@pytest.mark.parametrize("stage", ["preflight", "tiled_init", "first_frames", "headline"])
def test_a_rank_that_leaves_during_the_bring_up_costs_a_diagnostic_line_not_a_hang(stage):
    """The round-3 review's first item.  Rank 1 of 3 leaves the process at `stage` (PWN_BENCH_DIE_AT).  Whatever the others are
    stuck in then -- a gloo collective, the library's wait for a peer -- the run ends within the deadline with ONE JSON line
    from rank 0: "value": null, "incomplete": true, why, and per rank the stage it reached (rank 1's last mark included),
    and a non-zero exit status."""
    import time
    env = dict(os.environ)
    env.update(PWN_BENCH_ONE_DEVICE="1", PWN_BENCH_TRANSPORT="shm", MASTER_ADDR="127.0.0.1", PWN_BENCH_DIE_AT="%s:1" % stage)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "4", "--warmup", "1", "--min-time", "0.1", "--sweep-time", "0",
           "--bringup-timeout", "20", "--headline-timeout", "20", "--width", "1280", "--height", "720", "--no-d2h"]
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    took = time.time() - t0
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["value"] is None and d["incomplete"] is True and d["n_gpus"] == 3 and d["error"]
    st = {s["rank"]: s for s in d["stage_reached"]}
    assert set(st) == {0, 1, 2}
    assert st[1]["stage"] == stage or st[1]["stage"].endswith(stage), st[1]
    assert all(s["stage"] is not None for s in st.values())
    if stage != "preflight":
        assert len(d["tiling"]["preflight"]) == 3          # what was known before the loss is in the line
    assert took < 120, took


def test_bench_falls_back_when_rccl_does_not_come_up():
    """RCCL asked for with two ranks on ONE device -- a communicator that cannot be made here: every rank gets an
    error back (or librccl is missing altogether), the ranks agree on the shared-memory transport, and the line says
    what happened instead of the job dying without one."""
    env = dict(os.environ)
    env.update(PWN_BENCH_ONE_DEVICE="1", PWN_BENCH_TRANSPORT="rccl", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--min-time", "0.1",
           "--width", "1280", "--height", "720", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["frame_fnv64"] == "078fb94a5cd068f5"
    assert d["tiling"]["transport"] == "shm" and "transport_note" in d["tiling"], d["tiling"]


def test_bench_line_on_one_gpu():
    """The contract of the one-GPU line (the driver's BENCH run): one JSON line with the metric, the roofline object of
    the dominant kernel, the D2H-inclusive leg and the work counters; the last frame is the golden frame."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--min-time", "0.2",
           "--width", "1280", "--height", "720", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 2 and d["value"] > 0 and d["unit"] == "Mpixels/s"
    assert abs(d["value"] - 1280 * 720 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "model_over_measured", "valu_issue_frac_of_peak", "lane_slot_frac"):
        assert k in rf, k
    assert "issue_frac" not in rf
    assert d["vs_baseline"] is None           # (no published number for this metric: BASELINE.md)
    # the reference's own undefined corner, as the line states it: the three scenes with both reference renderings
    nf = d["parity"]["nonfinite_scenes"]
    assert len(nf) == 3 and all(q["equals_ieee_build"] is True and q["pixels_with_nonfinite_depth"] > 0 for q in nf), nf
    assert d["parity"]["headline_frame_equals_reference_golden"] is True
    assert "frames resident on the device" in d["metric"] and d["transport"] is None
    assert d["timing"]["roofline_leg"]["ms_per_step"] > 0 and d["timing"]["launches_timed"] > 0
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    assert abs(rf["achieved"] - 8 * 1280 * 720 / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 0.02 * rf["achieved"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["frame_fnv64"] == "078fb94a5cd068f5" and d["parity_vs_reference_golden"] is True
    assert d["d2h_inclusive"]["value"] > 0 and d["d2h_inclusive"]["last_frame_equals_resident_frame"] is True
    assert abs(d["work"]["steps_per_ray"] - 4.007) < 0.001 and d["work"]["rays_per_pixel"] == 3.0
