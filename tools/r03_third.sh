#!/bin/bash
OUT=gpurun_out/r03c
mkdir -p $OUT
export TMPDIR=/tmp
./tools/ba_bench > $OUT/ba_bench.txt 2>&1
cat $OUT/ba_bench.txt
./tools/microbench > $OUT/microbench.txt 2>&1
tail -8 $OUT/microbench.txt
timeout 600 python3 -m pytest tests/test_gpu_configs.py::test_rccl_process_group_of_one_rank -q -x 2>&1 | tail -3
