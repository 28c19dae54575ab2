# The randomised parity campaign that closes a round: generic and lattice scenes, the forced 4-lane variant, the
# refill scheduler, large frames -- new seeds, 17 240 scenes, ~14 GPU-minutes.
#   gpurun -- bash tools/fuzz_campaign.sh      -> gpurun_out/fuzz2/*.txt, one summary line per leg on stdout
mkdir -p gpurun_out/fuzz2
run() { tag=$1; shift; "$@" > gpurun_out/fuzz2/$tag.txt 2>&1; echo "$tag: $(tail -1 gpurun_out/fuzz2/$tag.txt)"; }
run g9101 python tools/fuzz_parity.py 3000 9101
run l9102 python tools/fuzz_parity.py 4000 9102 --lattice
run w9103 env PWN_DBG_FORCE_HASW=1 python tools/fuzz_parity.py 2000 9103
run wl9104 env PWN_DBG_FORCE_HASW=1 python tools/fuzz_parity.py 3000 9104 --lattice
run r9105 env PWN_SCHEDULER=refill python tools/fuzz_parity.py 2000 9105
run rl9106 env PWN_SCHEDULER=refill python tools/fuzz_parity.py 3000 9106 --lattice
run big9107 python tools/fuzz_parity.py 120 9107 --size 1920x1080
run bigl9108 python tools/fuzz_parity.py 120 9108 --lattice --size 1920x1080
