# The randomised parity campaign that closes a round: generic and lattice scenes, the forced 4-lane variant, the
# refill scheduler, large frames, and the frames in flight on two compute streams -- new seeds, 17 240 scenes + 420 frames + 14 tiled runs + 12 runs with a rank that leaves, ~18 GPU-minutes.
#   gpurun -- bash tools/fuzz_campaign.sh [SEED0]   -> gpurun_out/fuzz2/*.txt, one summary line per leg on stdout
#   (seeds SEED0 .. SEED0+7, default 9101)
S=${1:-9101}
mkdir -p gpurun_out/fuzz2
run() { tag=$1; shift; "$@" > gpurun_out/fuzz2/$tag.txt 2>&1; echo "$tag: $(tail -1 gpurun_out/fuzz2/$tag.txt)"; }
run g$((S+0)) python tools/fuzz_parity.py 3000 $((S+0))
run l$((S+1)) python tools/fuzz_parity.py 4000 $((S+1)) --lattice
run w$((S+2)) env PWN_DBG_FORCE_HASW=1 python tools/fuzz_parity.py 2000 $((S+2))
run wl$((S+3)) env PWN_DBG_FORCE_HASW=1 python tools/fuzz_parity.py 3000 $((S+3)) --lattice
run r$((S+4)) env PWN_SCHEDULER=refill python tools/fuzz_parity.py 2000 $((S+4))
run rl$((S+5)) env PWN_SCHEDULER=refill python tools/fuzz_parity.py 3000 $((S+5)) --lattice
run big$((S+6)) python tools/fuzz_parity.py 120 $((S+6)) --size 1920x1080
run bigl$((S+7)) python tools/fuzz_parity.py 120 $((S+7)) --lattice --size 1920x1080
run frames$((S+8)) python tools/fuzz_frames.py 300 $((S+8)) 1280x720
run frames4k$((S+9)) python tools/fuzz_frames.py 120 $((S+9)) 3840x2160
run tiled$((S+10)) python tools/fuzz_tiled.py 14 $((S+10))
run deadlines$((S+11)) python tools/fuzz_deadlines.py 12 $((S+11))
# round 5: the blocking call in row strips at every size (PWN_CALL_STRIPS: explicit counts ignore the size rule), both forms of the per-cell
# sphere lists forced, and the one-process group (tools/fuzz_group.py: 2..7 members on device 0 against one context)
run strips$((S+12)) env PWN_CALL_STRIPS=5 python tools/fuzz_parity.py 2000 $((S+12))
run stripsl$((S+13)) env PWN_CALL_STRIPS=3 python tools/fuzz_parity.py 2000 $((S+13)) --lattice
run stripsbig$((S+14)) env PWN_CALL_STRIPS=8 python tools/fuzz_parity.py 100 $((S+14)) --size 1920x1080
run inl$((S+15)) env PWN_SPHERE_LISTS=inline python tools/fuzz_parity.py 2000 $((S+15))
run inlw$((S+16)) env PWN_SPHERE_LISTS=inline PWN_DBG_FORCE_HASW=1 python tools/fuzz_parity.py 1500 $((S+16)) --lattice
run idx$((S+17)) env PWN_SPHERE_LISTS=indexed python tools/fuzz_parity.py 1500 $((S+17))
run group$((S+18)) python tools/fuzz_group.py 120 $((S+18))
