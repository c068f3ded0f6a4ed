#!/bin/bash
OUT=gpurun_out/r03e
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for P in 0 1; do
  SONIC_PROVE_PACK=$P timeout 900 python3 bench.py --no-cpu --msm-only 2>/dev/null > /dev/null
  SONIC_PROVE_PACK=$P timeout 900 python3 bench.py --no-cpu 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pack=$P prove streamed', j['ms_per_step'], 'seq', j['sequential']['ms_per_proof'], 'same', j['sequential']['same_bytes_as_streamed'], 'msm', j['msm']['ms_per_msm'], j['msm']['sequential']['ms_per_msm'])" | tee -a $OUT/pack.txt
done
done
for lg in 14 16 20; do
for P in 0 1; do
  SONIC_PROVE_PACK=$P timeout 900 python3 bench.py --no-cpu --log2n $lg --msm-log2 12 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg pack=$P prove streamed', j['ms_per_step'], 'seq', j['sequential']['ms_per_proof'])" | tee -a $OUT/pack.txt
done
done
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fs.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
