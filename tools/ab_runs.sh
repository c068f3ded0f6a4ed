for lg in 14 16 17 18; do
  for r in 1 0; do
    export SONIC_PROVE_RUNS=$r
    python3 bench.py --no-cpu --no-sensitivities --strong-log2n 0 --log2n $lg --msm-log2 12 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg runs=$r streamed %.2f  sequential %.2f unprepared %.2f one_shot %.2f' % (d['ms_per_step'], d['sequential']['ms_per_proof'], d['resident_unprepared']['ms_per_proof'], d['one_shot']['ms_per_proof']))"
  done
done
unset SONIC_PROVE_RUNS
python3 tools/throughput_mode.py 2>&1 | tail -1
SONIC_PROVE_RUNS=0 python3 tools/throughput_mode.py 2>&1 | tail -1
