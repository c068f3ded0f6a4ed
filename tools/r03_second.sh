#!/bin/bash
OUT=gpurun_out/r03b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -q --durations=20 > $OUT/pytest_gpu.txt 2>&1
tail -15 $OUT/pytest_gpu.txt
for K in 8 4 2 1; do
  SONIC_SLICE_SEGMENT=$K timeout 300 python3 bench.py --msm-strong --emulate-world 8 --no-cpu --steps 10 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j['msm_strong']['emulated_share']; print('K=$K', e['ms_per_share'], e['speedup_vs_single'], {k:v for k,v in e['kernel_ms'].items() if v>0.05})" >> $OUT/strong_K.txt 2>&1
done
cat $OUT/strong_K.txt
