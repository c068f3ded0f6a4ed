#!/bin/bash
OUT=gpurun_out/r03h
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for B in 256 128 64; do
  SONIC_ACCUM_BLOCK=$B timeout 600 python3 bench.py --no-cpu --steps 10 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['msm']['sequential']['kernel_ms']; print('block=$B prove streamed', j['ms_per_step'], 'seq', j['sequential']['ms_per_proof'], 'msm streamed', j['msm']['ms_per_msm'], 'seq', j['msm']['sequential']['ms_per_msm'], 'accum', k['k_bucket_accum'], 'strong', j['msm_strong']['ms_per_msm'])" | tee -a $OUT/block.txt
done
done
