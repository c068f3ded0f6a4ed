#!/bin/bash
OUT=gpurun_out/r03d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.txt 2>&1
tail -12 $OUT/pytest_gpu.txt
timeout 900 python3 bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err
python3 -c "
import json; j=json.load(open('$OUT/bench.json')); print('prove streamed', j['ms_per_step'], 'seq', j['sequential']['ms_per_proof'], 'msm streamed', j['msm']['ms_per_msm'], 'seq', j['msm']['sequential']['ms_per_msm'], j['msm']['sequential']['kernel_ms'])"
timeout 900 python3 bench.py --no-cpu > $OUT/bench2.json 2> $OUT/bench2.err
python3 -c "
import json; j=json.load(open('$OUT/bench2.json')); print('prove streamed', j['ms_per_step'], 'seq', j['sequential']['ms_per_proof'], 'msm streamed', j['msm']['ms_per_msm'], 'seq', j['msm']['sequential']['ms_per_msm'])"
