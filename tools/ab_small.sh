#!/bin/bash
# A/B of the small-proof knobs on ONE box: ms per proof streamed (two handles) / one at a time.  usage: bash tools/ab_small.sh "14 16" [steps]
export TMPDIR=/tmp
LGS=${1:-"14 16"}
STEPS=${2:-30}
run() {  # label, log2n, env...
  local label=$1 lg=$2; shift 2
  env "$@" python3 bench.py --log2n $lg --steps $STEPS --warmup 4 --no-cpu --prove-only --strong-log2n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg  %-34s streamed %6.2f  sequential %6.2f' % ('$label', d['ms_per_step'], d['sequential']['ms_per_proof']))"
}
for rep in 1 2; do
for lg in $LGS; do
  run "default" $lg A=1
  run "not fused (round 5 lanes)" $lg SONIC_PROVE_FUSED=0
  run "three streams per handle" $lg SONIC_FUSED_LANES=0
  run "stream priorities" $lg SONIC_PROVE_PRIORITIES=1
  run "t group as a chain of its own" $lg SONIC_FUSED_SPLIT_T=1
  if [ $lg -le 15 ]; then
    run "table c = 17" $lg SONIC_MSM_TABLE_C=17
    run "table c = 15" $lg SONIC_MSM_TABLE_C=15
  else
    run "table c = 16" $lg SONIC_MSM_TABLE_C=16
    run "C over the plain basis" $lg SONIC_PROVE_SYM=0
  fi
done
done
