#!/bin/bash
OUT=gpurun_out/r03j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -4
for R in 0 1; do
SONIC_NTT_RADIX8=$R timeout 900 python3 bench.py --no-cpu 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=j['roofline_ntt']; print('radix8=$R prove streamed', j['ms_per_step'], 'seq', j['sequential']['ms_per_proof'], 'ntt ms', n['ms_per_product'], n['frac'], n['hbm_passes_per_transform'], {k:v['ms_per_product'] for k,v in n['kernels'].items()})" | tee -a $OUT/ntt.txt
done
for lg in 16 20; do
SONIC_NTT_RADIX8=1 timeout 900 python3 bench.py --no-cpu --log2n $lg --msm-log2 12 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=j['roofline_ntt']; print('n=2^$lg prove streamed', j['ms_per_step'], 'seq', j['sequential']['ms_per_proof'], 'ntt ms', n['ms_per_product'], n['frac'], n['hbm_passes_per_transform'])" | tee -a $OUT/ntt.txt
done
