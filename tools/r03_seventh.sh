#!/bin/bash
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
for K in 8 4 2; do
  SONIC_MSM_SEGMENT=$K timeout 600 python3 bench.py --no-cpu --msm-only --steps 10 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['msm']['sequential']['kernel_ms']; print('K=$K msm streamed', j['msm']['ms_per_msm'], 'seq', j['msm']['sequential']['ms_per_msm'], 'segments', k['k_bucket_segments'], 'group', k['k_group_sum'], 'window', k['k_window_sum'])" | tee -a $OUT/seg_K.txt
done
for L in 16 8 4; do
  SONIC_PROVE_SEGMENT_LAST=$L timeout 600 python3 bench.py --no-cpu --prove-only 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LAST=$L prove streamed', j['ms_per_step'], 'seq', j['sequential']['ms_per_proof'])" | tee -a $OUT/seg_K.txt
done
rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $OUT/pmc_valu2 -o t -- python3 bench.py --no-cpu --no-pipeline --prove-only --steps 2 --warmup 1 > $OUT/pmc_valu.json 2> $OUT/pmc_valu.err
python3 tools/valu_budget.py $(find $OUT/pmc_valu2 -name "*counter_collection.csv" | head -1) 3 > $OUT/valu_budget.txt 2>&1
head -8 $OUT/valu_budget.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/solo2 -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pipeline --prove-only > $OUT/solo.json 2> $OUT/solo.err
python3 tools/timeline.py $(find $OUT/solo2 -name "*kernel_trace.csv" | head -1) 250 > $OUT/timeline_solo.txt 2>&1
head -3 $OUT/timeline_solo.txt
