#!/bin/bash
export TMPDIR=/tmp
run() {  # label, log2n, env...
  local label=$1 lg=$2; shift 2
  env "$@" python3 bench.py --log2n $lg --steps 30 --warmup 4 --no-cpu --prove-only --strong-log2n 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$lg  %-34s streamed %6.2f  sequential %6.2f' % ('$label', d['ms_per_step'], d['sequential']['ms_per_proof']))"
}
for rep in 1 2; do
for lg in 16 14 18; do
  for q in 8 12 16 24 32; do run "GPU_MAX_HW_QUEUES=$q" $lg GPU_MAX_HW_QUEUES=$q; done
done
done
