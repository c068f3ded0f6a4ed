export TMPDIR=/tmp
O=gpurun_out/r06_diag2; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for lg in 10 14 16; do
  python3 bench.py --no-cpu --no-sensitivities --strong-log2n 0 --log2n $lg --msm-log2 12 2>$O/b$lg.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg streamed %.2f  sequential %.2f' % (d['ms_per_step'], d['sequential']['ms_per_proof']))"
done
python3 tools/criterion_shape.py 2>&1 | tail -3
python3 tools/throughput_mode.py 2>&1 | tail -4
