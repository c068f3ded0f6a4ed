#!/bin/bash
OUT=gpurun_out/r03i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -q -x --durations=5 > $OUT/pytest_gpu.txt 2>&1
tail -9 $OUT/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 -c "
import json; j=json.load(open('$OUT/bench.json')); print('prove streamed', j['ms_per_step'], j['value'], 'seq', j['sequential']['ms_per_proof'], 'msm streamed', j['msm']['ms_per_msm'], 'seq', j['msm']['sequential']['ms_per_msm'], 'accum', j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['roofline']['rocprof'], j['int_roofline']['frac'], j['roofline_ntt']['frac'], j['cpu_baseline']['s_per_proof'], j['cpu_baseline']['same_bytes_as_gpu_proof'])"
