#!/bin/bash
# Runs of equal coefficients in the unprepared S_j through the running sums of the alpha basis: SONIC_PROVE_RUNS=1 / 0 by size, and the
# throughput of independent one-shot statements under the default rule / without.   bash tools/ab_runs.sh > profiles/<round>_runs_ab.txt
echo "# Runs of equal coefficients in the unprepared S_j through the running sums of the alpha basis (round 5): SONIC_PROVE_RUNS=1 / 0, ms per proof"
echo "# python3 bench.py --no-cpu --no-sensitivities --strong-log2n 0 --log2n <k> --msm-log2 12 under each value; streamed / sequential are the PREPARED handle (unaffected)"
for lg in 14 16 17 18; do
  for r in 1 0; do
    export SONIC_PROVE_RUNS=$r
    python3 bench.py --no-cpu --no-sensitivities --strong-log2n 0 --log2n $lg --msm-log2 12 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg runs=$r streamed %.2f  sequential %.2f unprepared %.2f one_shot %.2f' % (d['ms_per_step'], d['sequential']['ms_per_proof'], d['resident_unprepared']['ms_per_proof'], d['one_shot']['ms_per_proof']))"
  done
done
unset SONIC_PROVE_RUNS
python3 tools/throughput_mode.py 2>&1 | tail -1
SONIC_PROVE_RUNS=0 python3 tools/throughput_mode.py 2>&1 | tail -1
echo "# (first form, the small MSM behind the batch on the same lane: unprepared 7.28 / 13.61 / 21.25 / 34.61 at n = 2^14 / 16 / 17 / 18; on a stream of its own: 9.77 / 17.36 / 25.78 / 37.70 -- two more streams than hardware queues)"
