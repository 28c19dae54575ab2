#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_j; mkdir -p $O
{
for rep in 1 2; do bash tools/variants.sh "base head1" 3840 2160 40; done
bash tools/variants.sh "base head1" 1920 1080 60 synth64
bash tools/variants.sh "base head1" 7680 4320 20 synth256
} 2>&1 | grep -v amdgpu > $O/variants_head.txt
cat $O/variants_head.txt
for nn in 8 4 2; do STRIP_BALANCE=5 python tools/strip_time.py $nn 2>&1 | grep -v amdgpu > $O/strips_${nn}_balanced.txt; grep "slowest\|re-cut" $O/strips_${nn}_balanced.txt; done
STRIP_BALANCE=5 python tools/strip_time.py 8 7680 4320 2>&1 | grep -v amdgpu > $O/strips_8_8k_balanced.txt; grep "slowest\|re-cut" $O/strips_8_8k_balanced.txt
for rk in 0 3 4 7; do python tools/strip_wave_log.py 8 $rk 2>&1 | grep -v amdgpu; done > $O/strip_wave_logs.txt
python tools/wave_log.py 1280 720 2>&1 | grep -v amdgpu >> $O/strip_wave_logs.txt
python tools/wave_log.py 320 240 2>&1 | grep -v amdgpu >> $O/strip_wave_logs.txt
