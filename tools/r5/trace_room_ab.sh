cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do for room in -1 256 320 384 512; do
PWN_TRACE_ROOM=$room python3 bench.py --steps 50 --warmup 10 --min-time 1.5 --no-cpu-baseline --no-d2h 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('room %4s: two streams %.1f Mpix/s %.4f ms/frame | room_now %s' % ('$room', d['value'], d['ms_per_step'], d['config']['trace_room'].get('room_now')))"
done; done
