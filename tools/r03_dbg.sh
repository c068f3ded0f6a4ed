#!/bin/bash
OUT=gpurun_out/r03g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest "tests/test_gpu_configs.py::test_msm_2p22_sharded_on_one_gpu" -m gpu -q -x 2>&1 | tail -60 > $OUT/dbg1.txt
cat $OUT/dbg1.txt | grep -E "assert|Error|error|passed|failed" | head -20
