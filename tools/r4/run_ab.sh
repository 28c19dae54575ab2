# round 4, session AB: the line `bench.py --gpus 2` prints (two rank processes on ONE GPU over the shared-memory stand-in: structure, not speed)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_ab; mkdir -p $O
export PWN_BENCH_ONE_DEVICE=1 PWN_BENCH_TRANSPORT=shm MASTER_ADDR=127.0.0.1
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 10 --min-time 1 > $O/bench_2ranks_one_device_shm.json 2> $O/bench_2ranks.err
tail -c 400 $O/bench_2ranks.err
python3 - <<"PY"
import json
d=json.load(open("gpurun_out/r4_ab/bench_2ranks_one_device_shm.json"))
print(d["metric"][:160]); print(d["value"], d["ms_per_step"], d["tiling"]["choreography"][:40])
for k,v in d["tiling"]["sweep"].items(): print(k, v.get("value"), v.get("ms_per_step"))
print(d["d2h_inclusive"]["value"])
PY
