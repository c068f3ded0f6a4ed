#!/bin/bash
OUT=gpurun_out/r03a
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/cpu_scaling.py --log2 18 > $OUT/cpu_scaling.txt 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_dist.py "tests/test_gpu_parity.py::test_srs_refuses_points_at_infinity" "tests/test_gpu_parity.py::test_msm_entry_encoding_limits" "tests/test_gpu_parity.py::test_srs_file_round_trip" -q -x --durations=15 > $OUT/pytest_new.txt 2>&1
tail -5 $OUT/pytest_new.txt
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err; cat $OUT/bench.json | head -c 3000
timeout 600 python3 bench.py --msm-strong --emulate-world 8 --no-cpu --steps 10 > $OUT/strong8.json 2> $OUT/strong8.err
cat $OUT/strong8.json
