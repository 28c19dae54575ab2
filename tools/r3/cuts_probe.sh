#!/bin/bash
# 3 rank processes on one GPU, 4K, same scene, re-cut every 2 frames: what do the cost words look like?
cd "$GRAFT_REPO_ROOT"
ID=/tmp/pwn_cuts_probe_$$.id
rm -f $ID
for r in 0 1 2; do
  TILED_BALANCE=2 TILED_SAME_SCENE=1 python tools/tiled_rank.py $r 3 $ID shm 3840 2160 pwnfps_level 24 > /tmp/cuts_rank$r.txt 2>&1 &
done
wait
for r in 0 1 2; do echo "rank $r"; grep -E "^rows|^cuts" /tmp/cuts_rank$r.txt | tr '\n' ';'; echo; done
