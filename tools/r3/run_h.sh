#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_h; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt
tail -5 $O/tests.txt
{
for rep in 1 2; do
echo "== 4K level.txt"; bash tools/variants.sh "base exit1 exit2" 3840 2160 40
done
echo "== 1080p synth64"; bash tools/variants.sh "base exit1 exit2" 1920 1080 60 synth64
echo "== 8K synth256"; bash tools/variants.sh "base exit1 exit2" 7680 4320 20 synth256
echo "== 720p level.txt"; bash tools/variants.sh "base exit1 exit2" 1280 720 60
} > $O/variants.txt 2>&1
cat $O/variants.txt | grep -v amdgpu
python tools/region_counts.py $O/region_counts.json > $O/region_counts.log 2>&1; tail -3 $O/region_counts.log
python tools/strip_time.py 8 2>&1 | grep -v amdgpu > $O/strips8.txt; tail -1 $O/strips8.txt
