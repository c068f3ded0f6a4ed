#!/bin/bash
OUT=gpurun_out/r03f
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3; do
for V in alt main; do
  if [ $V = alt ]; then export SONIC_HIP_LIB=$PWD/tools/libsonic_hip_alt.so; else unset SONIC_HIP_LIB; fi
  timeout 900 python3 bench.py --no-cpu 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['msm']['sequential']['kernel_ms']; print('$V prove streamed', j['ms_per_step'], 'seq', j['sequential']['ms_per_proof'], 'msm', j['msm']['ms_per_msm'], j['msm']['sequential']['ms_per_msm'], 'segments', k['k_bucket_segments'], 'accum', k['k_bucket_accum'], 'strong', j['msm_strong']['ms_per_msm'])" | tee -a $OUT/seg.txt
done
done
unset SONIC_HIP_LIB
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -3
