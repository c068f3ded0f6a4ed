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
  run "stagger off" $lg SONIC_ACCUM_STAGGER=0
  run "priorities off" $lg SONIC_PROVE_PRIORITIES=0
  run "stagger off, priorities off" $lg SONIC_ACCUM_STAGGER=0 SONIC_PROVE_PRIORITIES=0
  run "accum block 64" $lg SONIC_FUSED_ACCUM_BLOCK=64
  run "accum block 64, priorities off" $lg SONIC_FUSED_ACCUM_BLOCK=64 SONIC_PROVE_PRIORITIES=0
  run "not fused (round 5 lanes)" $lg SONIC_PROVE_FUSED=0
  if [ $lg -le 15 ]; then
    run "table c = 17" $lg SONIC_MSM_TABLE_C=17
    run "table c = 16" $lg SONIC_MSM_TABLE_C=16
    run "table c = 14" $lg SONIC_MSM_TABLE_C=14
  else
    run "table c = 16" $lg SONIC_MSM_TABLE_C=16
    run "sym on" $lg SONIC_PROVE_SYM=1
  fi
done
done
