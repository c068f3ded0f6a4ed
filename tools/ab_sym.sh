#!/bin/bash
# C = commitPoly(s(u, Y)) over the SRS's symmetric sums (n terms instead of 2n + Q + 1): SONIC_PROVE_SYM=1 / 0 by size, ms per proof
echo "# C over the symmetric sums A[e] + A[-e] of the alpha basis: SONIC_PROVE_SYM=1 / 0, ms per proof (python3 bench.py --no-cpu --no-sensitivities --strong-log2n 0 --log2n <k> --msm-log2 12)"
for rep in 1 2; do
for lg in 14 16 18 20; do
  for r in 1 0; do
    export SONIC_PROVE_SYM=$r
    python3 bench.py --no-cpu --no-sensitivities --strong-log2n 0 --log2n $lg --msm-log2 12 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg sym=$r rep$rep streamed %.2f  sequential %.2f unprepared %.2f one_shot %.2f' % (d['ms_per_step'], d['sequential']['ms_per_proof'], d['resident_unprepared']['ms_per_proof'], d['one_shot']['ms_per_proof']))"
  done
done
done
