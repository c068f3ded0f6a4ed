#!/bin/bash
# A/B of hipGraph replay of a proof's enqueue (SONIC_PROVE_GRAPH=1) at the small sizes, where a proof is ~100 launches.  usage: bash tools/ab_graph.sh "10 12 14"
export TMPDIR=/tmp
LGS=${1:-"10 12 14"}
run() {
  local label=$1 lg=$2; shift 2
  env "$@" python3 bench.py --log2n $lg --steps 40 --warmup 6 --no-cpu --prove-only --strong-log2n 0 --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg  %-34s streamed %6.2f  sequential %6.2f' % ('$label', d['ms_per_step'], d['sequential']['ms_per_proof']))"
}
for rep in 1 2; do
for lg in $LGS; do
  run "launches" $lg A=1
  run "graph replay" $lg SONIC_PROVE_GRAPH=1
done
done
