# round 4, session W: the GPU suite again (bench-with-ranks test updated), then the N = 1 bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_w; mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $O/pytest_gpu.txt
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['d2h_inclusive']['value'])"
