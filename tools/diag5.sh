export TMPDIR=/tmp
O=gpurun_out/r06_diag5; mkdir -p $O
for lg in 10 13 14 15 16; do
  python3 bench.py --no-cpu --no-sensitivities --strong-log2n 0 --log2n $lg --msm-log2 12 2>$O/b$lg.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg streamed %.2f  sequential %.2f' % (d['ms_per_step'], d['sequential']['ms_per_proof']))"
done
echo "--- priorities off"
for lg in 14 16; do
  SONIC_PROVE_PRIORITIES=0 python3 bench.py --no-cpu --no-sensitivities --strong-log2n 0 --log2n $lg --msm-log2 12 2>$O/bp$lg.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg streamed %.2f  sequential %.2f' % (d['ms_per_step'], d['sequential']['ms_per_proof']))"
done
python3 tools/criterion_shape.py 2>&1 | tail -2
python3 tools/throughput_mode.py 2>&1 | tail -4
rocprofv3 --kernel-trace --output-format csv -d $O/str16 -o t -- python3 bench.py --log2n 16 --steps 8 --warmup 2 --no-cpu --prove-only --strong-log2n 0 > $O/str16.json 2> $O/str16.err
cp $(find $O/str16 -name "*kernel_trace.csv" | head -1) $O/str16_trace.csv; rm -rf $O/str16
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
