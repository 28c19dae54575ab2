# round 5: the lines `bench.py --gpus 2` (two rank processes) and `bench.py --gpus 2 --single-process` (one process, pwn_init_multi) print --
# both with their two ranks / members on ONE GPU (the shared-memory stand-in; copies between the members): structure, not speed
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_lines; mkdir -p $O
export PWN_BENCH_ONE_DEVICE=1 PWN_BENCH_TRANSPORT=shm MASTER_ADDR=127.0.0.1
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 50 --warmup 10 --min-time 1 > $O/bench_2ranks_one_device_shm.json 2> $O/bench_2ranks.err
tail -c 300 $O/bench_2ranks.err
python3 bench.py --gpus 2 --single-process --steps 50 --warmup 10 --min-time 1 > $O/bench_2members_one_device.json 2> $O/bench_2members.err
tail -c 300 $O/bench_2members.err
python3 - <<"PY"
import json
d=json.load(open("gpurun_out/r5_lines/bench_2ranks_one_device_shm.json"))
t=d["tiling"]
print(d["metric"][:150]); print("value", d["value"], d["ms_per_step"])
print("first_legs", {k:(v.get("value"), v.get("predicted_ms_per_frame") or v.get("link_model")) for k,v in t.get("first_legs",{}).items()})
print("best", t.get("best")); print("link_model", t.get("link_model"))
print("sweep", {k:v.get("value") for k,v in t.get("sweep",{}).items()})
g=json.load(open("gpurun_out/r5_lines/bench_2members_one_device.json"))
print(g["metric"][:150]); print("value", g["value"], g["ms_per_step"], g.get("transport"), g.get("transport_note"))
PY
