#!/bin/bash
# one bench.py line per BASELINE configuration on one GPU (resident loop and, for the sizes that matter, the D2H leg)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/configs
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
w=d['work']; k=d['kernel_ms']; p=d.get('d2h_inclusive') or {}
print('%-13s %4dx%-4d | %9.1f Mpix/s  %.4f ms/frame | trace %.4f blur %.4f | steps/ray %.3f lanes %.3f residency %.3f | d2h %s | one blocking call %s | fnv %s' % (
  d['config']['level'], d['config']['width'], d['config']['height'], d['value'], d['ms_per_step'], k['trace'], k['blur'], w['steps_per_ray'], w['walk_active_lane_fraction'], w['mean_wave_residency'], p.get('value'), p.get('blocking_call_mpix_s'), d['frame_fnv64']))"; }
run --level pwnfps_level --width 320 --height 240
run --level pwnfps_level --width 1280 --height 720
run --level synth64 --width 1920 --height 1080
run --level pwnfps_level --width 3840 --height 2160
run --level synth256 --width 7680 --height 4320
run --level pwnfps_level --width 7680 --height 4320
